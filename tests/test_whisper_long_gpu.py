"""-m gpu: Whisper decodes at the LENGTHS the bench and the reference run, against oracle/whisper_ref.py.

The reference feeds the previous segment's transcript as `initial_prompt` (/root/reference/back/api.py:1424-1426, 1467-1468), so
a `transcribe` call (1286-1292, 1432-1438, 1474-1480) decodes from up to 1 + 223 + 1 = 225 initial tokens and samples up to
224 more: 448 positions = n_text_ctx.  bench.py's pinned decodes run from the fixed prompts (10 tokens) to 224 sampled tokens
(~235 positions).  In `dec_attention_kernel<true>` (csrc/decoder.hip) wave w owns the self-attention keys [64 w, 64 w + 64) and
the loop strides by 256, so only decodes past position 64 use waves 1-3 and only decodes past 256 the second key round; the learned
positional embedding, ApplyTimestampRules over long histories and the n_text_ctx edge are exercised by nothing shorter either.

Every GPU token is walked through the oracle's KV-cached decoder (the arithmetic of `decoder_logits`, one token per call) under
teacher forcing: it must be an eps-argmax of the oracle's filtered logits (eps 0.02 at mini dims, 0.0275 at full small.en size: 2.5 x
the worst shortfall measured, 7.9e-3 / 1.1e-2; the short-decode tests use 0.05 / 0.08), equal to the argmax where the oracle's margin
exceeds 2 eps, and the summed log-probability and the no-speech probability must agree.  Paths: the split-KV kernels of <= 16 sequences ("kv16"), the lean K / V stream of 17 - 80
sequences ("kv_stream", the default of that range) and the cross attention against the encoder output ("xa_stream", > 80
sequences and every group of bench.py).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import within

from clearconverse_amd import _lib
from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.tokenizer import DecodeRules
from clearconverse_amd.weights import WhisperDims, synthetic_whisper_state_dict
from oracle import whisper_ref as R

pytestmark = pytest.mark.gpu


def _rules():
    r = DecodeRules()
    return r, R.Rules(suppress=tuple(r.suppress))


def _clips(lengths_s, seed0=0):
    clips = [synthetic_clip(seed0 + i, 30.0)[: int(s * 16000)] for i, s in enumerate(lengths_s)]
    n = [len(c) for c in clips]
    host = np.zeros((len(clips), max(n)), dtype=np.float32)
    for i, c in enumerate(clips):
        host[i, : len(c)] = c
    return n, torch.from_numpy(host).cuda()


def _long_prompt(rules, n_prompt, seed):
    """[sot_prev] + n_prompt text ids + [sot]: what transcribe() builds from a previous transcript (decoding.py::_get_initial_tokens
    keeps the last n_text_ctx // 2 - 1 = 223 prompt tokens)."""
    g = np.random.default_rng(seed)
    return [rules.sot_prev] + [int(x) for x in g.integers(1000, 40000, n_prompt)] + [rules.sot]


def walk_cached(orc, xa_row, prompt, result, sample_len, tol, name):
    """Teacher-forced walk of a GPU decode through oracle/whisper_ref.CachedDecoder.  Returns the number of steps whose oracle margin
    was decisive (the GPU token had to equal the oracle's argmax there)."""
    rules, orules = _rules()
    toks = result["tokens"]
    forced = toks + ([rules.eot] if len(toks) < sample_len else [])
    dec = R.CachedDecoder(orc, xa_row)
    logits = dec.step(torch.tensor([prompt], dtype=torch.long))[0]
    no_speech = float(F.softmax(logits[len(prompt) - 1].float(), dim=-1)[orules.no_speech])
    last, sampled, slp, decisive, worst = logits[-1], [], 0.0, 0, 0.0
    for i, t in enumerate(forced):
        lg = R.apply_filters(last, sampled, orules)
        top2 = torch.topk(lg, 2).values
        short = float(top2[0] - lg[t])
        assert short <= tol, (name, "step", i, "position", len(prompt) + i, t, int(lg.argmax()), short)
        worst = max(worst, short)
        if float(top2[0] - top2[1]) > 2 * tol:
            assert t == int(lg.argmax()), (name, i, t, int(lg.argmax()))
            decisive += 1
        slp += float(F.log_softmax(lg.float(), dim=-1)[t])
        sampled.append(t)
        if i + 1 < len(forced):
            last = dec.step(torch.tensor([[t]], dtype=torch.long))[0, -1]
    within(f"{name}: worst shortfall of a GPU token below the oracle's best filtered logit (teacher forced, <= 224 steps)", worst, tol)
    within(f"{name}: |sum_logprob - oracle (teacher forced)| / max(1, |oracle|)", abs(slp - result["sum_logprob"]) / max(1.0, abs(slp)), 2e-3)
    within(f"{name}: |no_speech_prob - oracle|", abs(no_speech - result["no_speech_prob"]), 1e-6 + 2e-3 * no_speech)
    return len(forced), decisive


PATHS = {"kv16": dict(B=2, env=None), "kv_stream": dict(B=24, env=None), "xa_stream": dict(B=24, env="1"), "xa_stream_b2": dict(B=2, env="1")}


@pytest.mark.parametrize("path", list(PATHS))
def test_mini_decodes_of_224_tokens_and_the_448_position_edge(ccx_ctx, monkeypatch, path):
    """Mini dims (2 layers of 128: the kernels of the full model on fewer, narrower layers), sample_len 224: sequence 0 from the
    one-token prompt [sot] (positions 0 .. 224), sequence 1 from a 225-token prompt (positions 0 .. 447: the n_text_ctx edge; the prompt is
    prefilled in 15 passes of 16 positions).  B = 24 decodes the two windows twelve times: copies must be bit-identical."""
    from clearconverse_amd.whisper import WhisperModel
    cfg = PATHS[path]
    dims = WhisperDims.mini(n_layer=2, n_state=128)
    sd = synthetic_whisper_state_dict(dims, seed=3)
    m = WhisperModel(dims, sd, max_batch=24, ctx=ccx_ctx)
    try:
        rules, _ = _rules()
        n, dev = _clips([6.0, 11.0])
        reps = cfg["B"] // 2
        big = dev.repeat(reps, 1).contiguous()
        m.log_mel(big, n * reps)
        xa = m.encode(cfg["B"], return_xa=True).cpu()
        prompts = [[rules.sot], _long_prompt(rules, 223, 1)]
        assert len(prompts[1]) == 225 and len(prompts[1]) + 224 - 1 == dims.n_text_ctx
        if cfg["env"]:
            monkeypatch.setenv("CCX_CROSS_X_MIN_ROWS", cfg["env"])
        res = m.decode_greedy(prompts * reps, sample_len=224)
        assert m.last_cross_path == path.replace("_b2", ""), m.last_cross_path
        for i in range(cfg["B"]):
            assert res[i]["tokens"] == res[i % 2]["tokens"] and res[i]["sum_logprob"] == res[i % 2]["sum_logprob"], i
        orc = R.WhisperRef(R.Dims(**dims.__dict__), sd)
        steps = decisive = 0
        for i in range(2):
            assert len(res[i]["tokens"]) >= 200, (i, len(res[i]["tokens"]))        # random weights: eot is one id of 51864
            a, c = walk_cached(orc, xa[i:i + 1], prompts[i], res[i], 224, 0.02, f"whisper mini long decode [{path}]")
            steps += a; decisive += c
        assert decisive >= 20, (steps, decisive)
        # stepwise prompt feeding (CCX_PREFILL=0) runs positions 0 .. 447 as 448 decode steps: same kernels per row
        monkeypatch.setenv("CCX_PREFILL", "0")
        stepwise = m.decode_greedy(prompts * reps, sample_len=224)
        monkeypatch.delenv("CCX_PREFILL")
        for i in range(2):
            if path == "kv_stream":
                # 17 - 80 sequences: the prefill's cross attention is its own kernel (dec_cross_prefill_kernel, four prompt rows per
                # block) and sums a row's keys in another order than the step kernel (dec_cross_stream_kernel): equal to rounding
                if stepwise[i]["tokens"] != res[i]["tokens"]:
                    walk_cached(orc, xa[i:i + 1], prompts[i], stepwise[i], 224, 0.02, f"whisper mini long decode, stepwise prompt [{path}]")
                else:
                    within("whisper mini long decode [kv_stream]: |sum_logprob prefilled - stepwise prompt| / max(1, |.|)",
                           abs(stepwise[i]["sum_logprob"] - res[i]["sum_logprob"]) / max(1.0, abs(res[i]["sum_logprob"])), 3.5e-5, i)
            else:
                assert stepwise[i]["tokens"] == res[i]["tokens"] and stepwise[i]["sum_logprob"] == res[i]["sum_logprob"], i
        # the edge: one more position is refused, loudly
        with pytest.raises(_lib.CcxError):
            m.decode_greedy([_long_prompt(rules, 224, 2)] * 2, sample_len=224)
        with pytest.raises(_lib.CcxError):
            m.decode_greedy(prompts, sample_len=225)
    finally:
        m.close()


@pytest.mark.parametrize("path", ["kv16", "xa_stream_b2"])
def test_full_size_decodes_of_224_tokens_against_the_oracle(ccx_ctx, monkeypatch, path):
    """Full small.en (12 + 12 layers of 768, 12 heads), two sequences, 224 sampled tokens each -- from [sot] and from a 225-token
    prompt (448 positions) -- teacher-forced through the oracle's cached decoder (seconds)."""
    from clearconverse_amd.whisper import WhisperModel
    dims = WhisperDims.small_en()
    sd = synthetic_whisper_state_dict(dims, seed=0)
    m = WhisperModel(dims, sd, max_batch=2, ctx=ccx_ctx)
    try:
        rules, _ = _rules()
        n, dev = _clips([30.0, 9.0])
        m.log_mel(dev, n)
        xa = m.encode(2, return_xa=True).cpu()
        prompts = [[rules.sot], _long_prompt(rules, 223, 3)]
        if path != "kv16":
            monkeypatch.setenv("CCX_CROSS_X_MIN_ROWS", "1")
        res = m.decode_greedy(prompts, sample_len=224)
        assert m.last_cross_path == path.replace("_b2", "")
        orc = R.WhisperRef(R.Dims(**dims.__dict__), sd)
        decisive = 0
        for i in range(2):
            assert len(res[i]["tokens"]) >= 200
            decisive += walk_cached(orc, xa[i:i + 1], prompts[i], res[i], 224, 0.0275, f"whisper small.en FULL size long decode [{path}]")[1]
        assert decisive >= 20, decisive
    finally:
        m.close()


def test_full_size_default_path_of_24_sequences_against_the_oracle_and_the_xa_stream(ccx_ctx, monkeypatch):
    """The default path of 17 - 80 sequences (per-layer K / V through `dec_cross_stream_kernel`, csrc/decoder.hip) at full small.en
    size: 24 sequences (6 windows x 4 prompt shapes), default environment, 48 sampled tokens -- every sequence an eps-argmax string
    of the oracle, and against the same windows on the X-stream path: equal tokens and log-probabilities to rounding, or two
    eps-argmax strings where a near-tie tips."""
    from clearconverse_amd.whisper import WhisperModel
    dims = WhisperDims.small_en()
    sd = synthetic_whisper_state_dict(dims, seed=0)
    m = WhisperModel(dims, sd, max_batch=24, ctx=ccx_ctx)
    try:
        rules, _ = _rules()
        n, dev = _clips([30.0, 9.0, 4.0, 17.5, 2.0, 24.0])
        big = dev.repeat(4, 1).contiguous()
        m.log_mel(big, n * 4)
        xa = m.encode(24, return_xa=True).cpu()
        shapes = [[rules.sot], [rules.sot_prev, 1212, 318, rules.sot], _long_prompt(rules, 38, 5), _long_prompt(rules, 15, 6)]
        prompts = [shapes[i // 6] for i in range(24)]
        S = 48
        res = m.decode_greedy(prompts, sample_len=S)
        assert m.last_cross_path == "kv_stream"
        monkeypatch.setenv("CCX_CROSS_X_MIN_ROWS", "1")
        xs = m.decode_greedy(prompts, sample_len=S)
        assert m.last_cross_path == "xa_stream"
        monkeypatch.delenv("CCX_CROSS_X_MIN_ROWS")
        orc = R.WhisperRef(R.Dims(**dims.__dict__), sd)
        tipped = 0
        for i in range(24):
            walk_cached(orc, xa[i:i + 1], prompts[i], res[i], S, 0.0275, "whisper small.en FULL size, 24 sequences [kv_stream]")
            if res[i]["tokens"] == xs[i]["tokens"]:
                within("whisper small.en FULL size: |sum_logprob kv_stream - xa_stream| / max(1, |.|)",
                       abs(res[i]["sum_logprob"] - xs[i]["sum_logprob"]) / max(1.0, abs(xs[i]["sum_logprob"])), 8e-4, i)
            else:
                tipped += 1
                walk_cached(orc, xa[i:i + 1], prompts[i], xs[i], S, 0.0275, "whisper small.en FULL size, 24 sequences [xa_stream]")
        assert tipped <= 6, tipped
    finally:
        m.close()
