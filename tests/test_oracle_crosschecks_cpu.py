"""CPU: the unpinned oracles against INDEPENDENT library code that is importable in this image.

The reference holds no fixture for any numerical model and none of its dependencies (pyannote.audio, speechbrain,
openai-whisper) is installed, so oracle/pyannote_ref.py, sepformer_ref.py and whisper_ref.py stay "parity unpinned"
(DESIGN.md section 3).  What CAN be checked here is that their hand-written building blocks equal the library modules the
upstream packages themselves are built from, with shared weights:
  * PyanNet's recurrent core is `torch.nn.LSTM(60, 128, num_layers=4, bidirectional=True, batch_first=True)`;
  * SpeechBrain's TransformerEncoderLayer wraps `torch.nn.MultiheadAttention` + `nn.LayerNorm` (pre-norm);
  * openai-whisper's SuppressBlank / SuppressTokens / ApplyTimestampRules and greedy loop were re-implemented independently in
    `transformers` (`SuppressTokensAtBeginLogitsProcessor`, `SuppressTokensLogitsProcessor`, `WhisperTimeStampLogitsProcessor`,
    `WhisperForConditionalGeneration.generate`).
A shared misreading of the upstream ARCHITECTURE (layer sizes, kernel widths, the order of blocks) is still invisible to these
tests; a slip in the recurrence, the attention arithmetic or the decoding rules is not."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.tokenizer import BLANK, EOT, NO_TIMESTAMPS, SUPPRESS_TOKENS, TIMESTAMP_BEGIN, DecodeRules
from clearconverse_amd.weights import SepDims, WhisperDims, synthetic_pyannet_state_dict, synthetic_sepformer_state_dict, synthetic_whisper_state_dict
from oracle import pyannote_ref as P
from oracle import sepformer_ref as S
from oracle import whisper_ref as R


# ------------------------------------------------------------------------------------------------ PyanNet recurrence
def test_pyannet_lstm_loop_equals_torch_nn_lstm():
    """oracle/pyannote_ref.py:118-140 (hand-rolled 4-layer BiLSTM, gate order i,f,g,o, reversed direction by flipping)
    vs torch.nn.LSTM with the same state dict -- through the whole network (SincNet front end and heads shared)."""
    sd = dict(synthetic_pyannet_state_dict(7, seed=11)); sd["powerset"] = torch.tensor(1)
    lstm = torch.nn.LSTM(60, 128, num_layers=4, bidirectional=True, batch_first=True)
    lstm.load_state_dict({k[len("lstm."):]: v for k, v in sd.items() if k.startswith("lstm.")}, strict=True)
    wav = torch.from_numpy(synthetic_clip(4, 10.0)[:48000].copy())[None, None]
    with torch.no_grad():
        ours = P.pyannet_forward(sd, wav)
        x = P.sincnet_forward(sd, wav).permute(0, 2, 1)
        x, _ = lstm(x)
        for i in range(2):
            x = F.leaky_relu(F.linear(x, sd[f"linear.{i}.weight"], sd[f"linear.{i}.bias"]))
        ref = F.log_softmax(F.linear(x, sd["classifier.weight"], sd["classifier.bias"]), dim=-1)
    assert ours.shape == ref.shape
    assert float((ours - ref).abs().max()) < 2e-5
    assert torch.equal(ours.argmax(-1), ref.argmax(-1))


# ------------------------------------------------------------------------------------------------ SepFormer layer
def test_sepformer_block_equals_torch_transformer_modules():
    """oracle/sepformer_ref.py::_block (pre-norm layers with a hand-written packed-QKV attention, final LayerNorm, gLN, skip)
    vs torch.nn.TransformerEncoderLayer(norm_first=True, activation=relu) -- which is nn.MultiheadAttention + nn.LayerNorm +
    two Linears, the modules SpeechBrain's TransformerEncoderLayer wraps -- with shared weights."""
    dims = SepDims(n_layers=2, n_blocks=1)
    sd = synthetic_sepformer_state_dict(dims, seed=9)
    ref = S.SepformerRef(S.SepDims(**dims.__dict__), sd)
    prefix = "masknet.model.seg_model.0"
    D = dims.d_model
    layers = []
    for l in range(dims.n_layers):
        p = f"{prefix}.mdl.layers.{l}"
        lay = torch.nn.TransformerEncoderLayer(D, dims.n_head, dims.d_ffn, dropout=0.0, activation="relu", layer_norm_eps=1e-6,
                                               batch_first=True, norm_first=True)
        lay.load_state_dict({
            "self_attn.in_proj_weight": sd[p + ".self_att.att.in_proj_weight"], "self_attn.in_proj_bias": sd[p + ".self_att.att.in_proj_bias"],
            "self_attn.out_proj.weight": sd[p + ".self_att.att.out_proj.weight"], "self_attn.out_proj.bias": sd[p + ".self_att.att.out_proj.bias"],
            "linear1.weight": sd[p + ".pos_ffn.ffn.0.weight"], "linear1.bias": sd[p + ".pos_ffn.ffn.0.bias"],
            "linear2.weight": sd[p + ".pos_ffn.ffn.3.weight"], "linear2.bias": sd[p + ".pos_ffn.ffn.3.bias"],
            "norm1.weight": sd[p + ".norm1.norm.weight"], "norm1.bias": sd[p + ".norm1.norm.bias"],
            "norm2.weight": sd[p + ".norm2.norm.weight"], "norm2.bias": sd[p + ".norm2.norm.bias"]}, strict=True)
        layers.append(lay.eval())
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, dims.segment, D, generator=g)
    with torch.no_grad():
        ours = ref._block(prefix, x)
        h = x + S.positional_encoding(dims.segment, D)
        for lay in layers:
            h = lay(h)
        h = F.layer_norm(h, (D,), sd[prefix + ".mdl.norm.norm.weight"], sd[prefix + ".mdl.norm.norm.bias"], 1e-6)
        # GlobalLayerNorm == GroupNorm with ONE group over (channel, time) and a per-channel affine, eps 1e-8
        h = F.group_norm(h.transpose(1, 2), 1, sd[prefix + ".norm.weight"].flatten(), sd[prefix + ".norm.bias"].flatten(), eps=1e-8).transpose(1, 2)
        want = h + x
    assert float((ours - want).abs().max()) < 5e-5 * max(1.0, float(want.abs().max()))


def test_sepformer_positional_encoding_is_the_standard_interleaved_table():
    pe = S.positional_encoding(150, 128)
    pos = np.arange(150)[:, None]
    i = np.arange(64)[None, :]
    ang = pos / np.power(10000.0, 2 * i / 128.0)
    assert np.abs(pe[:, 0::2].numpy() - np.sin(ang)).max() < 1e-5 and np.abs(pe[:, 1::2].numpy() - np.cos(ang)).max() < 2e-5


# ------------------------------------------------------------------------------------------------ Whisper decoding rules
class _GenCfg:
    """The attributes WhisperTimeStampLogitsProcessor reads from a GenerationConfig."""
    no_timestamps_token_id = NO_TIMESTAMPS
    eos_token_id = EOT
    bos_token_id = EOT
    max_initial_timestamp_index = 50
    _detect_timestamp_from_logprob = True


def _hf_filters(prompt_len):
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                        WhisperTimeStampLogitsProcessor)
    return [SuppressTokensAtBeginLogitsProcessor([BLANK, EOT], begin_index=prompt_len),          # SuppressBlank
            SuppressTokensLogitsProcessor(list(SUPPRESS_TOKENS)),                                 # SuppressTokens
            WhisperTimeStampLogitsProcessor(_GenCfg(), begin_index=prompt_len)]                   # ApplyTimestampRules


def _random_history(rng, n):
    """A plausible sampled prefix: text runs and timestamp singles / pairs, non-decreasing timestamps."""
    out, ts = [], TIMESTAMP_BEGIN + int(rng.integers(0, 40))
    while len(out) < n:
        kind = rng.integers(0, 3)
        if kind == 0:
            out.append(int(rng.integers(0, 50000)))
        elif kind == 1:
            out.append(ts)
        else:
            out += [ts, ts]
        ts += int(rng.integers(0, 30))
    return out[:n]


def test_apply_filters_equals_transformers_processors_on_random_logits():
    """oracle/whisper_ref.py::apply_filters vs the three transformers processors chained in openai-whisper's order, on random
    logits after random histories (0 .. 12 sampled tokens, every ApplyTimestampRules state).  The -inf pattern must be identical
    and the surviving logits untouched."""
    rng = np.random.default_rng(0)
    rules = R.Rules(suppress=tuple(SUPPRESS_TOKENS))
    prompt = [50360, 1212, 318, 50257]
    procs = _hf_filters(len(prompt))
    states = set()
    for trial in range(160):
        n = int(rng.integers(0, 13)) if trial >= 8 else 0
        hist = _random_history(rng, n)
        lg = torch.from_numpy(rng.normal(0, 3, 51864).astype(np.float32))
        if trial % 3 == 0:                       # make the timestamp mass win sometimes (the "force timestamps" rule)
            lg[TIMESTAMP_BEGIN:] += 6.0
        ours = R.apply_filters(lg, hist, rules)
        ids = torch.tensor([prompt + hist])
        sc = lg[None].clone()
        for pr in procs:
            sc = pr(ids, sc)
        assert torch.equal(torch.isinf(ours), torch.isinf(sc[0])), (trial, hist)
        keep = ~torch.isinf(ours)
        assert torch.equal(ours[keep], sc[0][keep])
        assert int(ours.argmax()) == int(sc[0].argmax())
        states.add((len(hist) == 0, len(hist) >= 1 and hist[-1] >= TIMESTAMP_BEGIN, len(hist) < 2 or hist[-2] >= TIMESTAMP_BEGIN))
    assert len(states) >= 5                      # first step, text/text, text/ts, ts/ts, ts/text all seen


def test_greedy_decode_equals_transformers_generate_with_timestamps():
    """Free-running greedy decode of the oracle vs `WhisperForConditionalGeneration.generate` (transformers' own sampling loop,
    KV cache and processors) on the shared seeded mini model: identical token ids.  Known rule difference: none for a single
    30 s window at temperature 0 -- HF's generate adds its own long-form / fallback logic only for longer inputs."""
    from tests.test_oracle_whisper import _hf_model
    dims = WhisperDims.mini(n_layer=2, n_state=128)
    sd = synthetic_whisper_state_dict(dims, seed=5)
    orc = R.WhisperRef(R.Dims(**dims.__dict__), sd)
    hf = _hf_model(dims, sd)
    rules = R.Rules(suppress=tuple(SUPPRESS_TOKENS))
    n_new = 12
    for ci, prompt in ((2, [50257]), (3, [50360, 1212, 318, 257, 50257])):
        clip = synthetic_clip(ci, 30.0)[: 16000 * 6]
        mel = R.pad_or_trim(R.log_mel_spectrogram(torch.from_numpy(clip))[:, : len(clip) // 160], 3000)[None]
        with torch.no_grad():
            xa = orc.encode(mel)
            ours = R.greedy_decode(orc, xa, [prompt], rules, sample_len=n_new)[0]
            # transformers' own greedy loop (sample.GreedySearch + DynamicCache) with the same three processors
            from transformers import LogitsProcessorList
            procs = LogitsProcessorList(_hf_filters(len(prompt)))
            out = hf.generate(input_features=mel, decoder_input_ids=torch.tensor([prompt]), logits_processor=procs, max_new_tokens=n_new,
                              do_sample=False, num_beams=1, eos_token_id=EOT, pad_token_id=EOT, return_timestamps=False,
                              suppress_tokens=None, begin_suppress_tokens=None, forced_decoder_ids=None)
        got = out[0, len(prompt):].tolist() if out.shape[1] > n_new else out[0].tolist()
        got = got[: got.index(EOT)] if EOT in got else got
        want = ours.tokens
        # decisive steps only: where the oracle's top-2 margin is below fp32 noise the two fp32 implementations may differ
        k = next((i + 1 for i, m in enumerate(ours.margins) if m < 1e-4), len(want))
        assert got[:k] == want[:k] and k >= min(5, len(want)), (got, want, ours.margins)
