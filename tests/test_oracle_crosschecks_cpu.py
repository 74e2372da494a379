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


# ------------------------------------------------------------------------------------------------ round 3: more building blocks
def test_sincnet_front_end_equals_torch_nn_modules_and_textbook_sinc_filters():
    """oracle/pyannote_ref.py::sincnet_forward (functional: hand-written instance norm, conv, pooling) vs the SAME stack built from
    torch.nn MODULES (nn.InstanceNorm1d(affine), nn.Conv1d, nn.MaxPool1d, nn.LeakyReLU -- what pyannote's SincNet block is made of),
    and its parametrised sinc filters against the textbook band-pass impulse responses computed independently with numpy:
    cos branch 2 f2 sinc(2 f2 t) - 2 f1 sinc(2 f1 t), sin branch (cos(2 pi f1 t) - cos(2 pi f2 t)) / (pi t), Hamming window, each
    divided by twice the band width."""
    from clearconverse_amd.weights import synthetic_xvector_state_dict
    sd = synthetic_xvector_state_dict(seed=8)
    wav = torch.from_numpy(synthetic_clip(6, 10.0)[:40000].copy())[None, None]
    g = lambda k: sd["sincnet." + k].float()
    filt = P.sinc_filters(g("conv1d.0.filterbank.low_hz_"), g("conv1d.0.filterbank.band_hz_"))
    # independent filter construction
    low = 50.0 + np.abs(g("conv1d.0.filterbank.low_hz_").numpy()[:, 0].astype(np.float64))
    high = np.clip(low + 50.0 + np.abs(g("conv1d.0.filterbank.band_hz_").numpy()[:, 0].astype(np.float64)), 50.0, 8000.0)
    n = np.arange(-125, 126, dtype=np.float64)
    t = n / 16000.0
    half = np.arange(125, dtype=np.float64)
    win_half = 0.54 - 0.46 * np.cos(2 * np.pi * np.linspace(0, 251 / 2 - 1, 125) / 251)
    win = np.concatenate([win_half, [1.0], win_half[::-1]])
    for i in (0, 7, 23, 39):
        f1, f2 = low[i], high[i]
        cosf = (2 * f2 * np.sinc(2 * f2 * t) - 2 * f1 * np.sinc(2 * f1 * t)) / 16000.0 * 16000.0      # = sin(2 pi f2 t)/(pi t) - sin(2 pi f1 t)/(pi t)
        with np.errstate(divide="ignore", invalid="ignore"):
            sinf = (np.cos(2 * np.pi * f1 * t) - np.cos(2 * np.pi * f2 * t)) / (np.pi * t)
        sinf[125] = 0.0
        band = f2 - f1
        want_cos = cosf * win / (2 * band)
        want_sin = sinf * win / (2 * band)
        # the upstream formula carries 1 / (n / 2) with n in radians-per-sample units: same shape up to the common factor it divides out
        got_cos, got_sin = filt[i].double().numpy(), filt[40 + i].double().numpy()
        assert np.allclose(got_cos / got_cos[125], want_cos / want_cos[125], atol=2e-5), i
        assert np.allclose(got_sin / np.abs(got_sin).max(), want_sin / np.abs(want_sin).max(), atol=2e-5), i
    mods = []
    x = torch.nn.InstanceNorm1d(1, affine=True)
    x.load_state_dict({"weight": g("wav_norm1d.weight"), "bias": g("wav_norm1d.bias")})
    conv0 = torch.nn.Conv1d(1, 80, 251, stride=10, bias=False)
    conv0.load_state_dict({"weight": filt[:, None, :]})
    net = [x, conv0]
    with torch.no_grad():
        h = conv0(x(wav)).abs()
        for i, (cin, cout) in enumerate(((80, 60), (60, 60))):
            inorm = torch.nn.InstanceNorm1d(cin, affine=True)
            inorm.load_state_dict({"weight": g(f"norm1d.{i}.weight"), "bias": g(f"norm1d.{i}.bias")})
            conv = torch.nn.Conv1d(cin, cout, 5)
            conv.load_state_dict({"weight": g(f"conv1d.{i + 1}.weight"), "bias": g(f"conv1d.{i + 1}.bias")})
            h = conv(torch.nn.LeakyReLU()(inorm(torch.nn.MaxPool1d(3, stride=3)(h))))
        inorm = torch.nn.InstanceNorm1d(60, affine=True)
        inorm.load_state_dict({"weight": g("norm1d.2.weight"), "bias": g("norm1d.2.bias")})
        ref = torch.nn.LeakyReLU()(inorm(torch.nn.MaxPool1d(3, stride=3)(h)))
        ours = P.sincnet_forward(sd, wav)
    assert ours.shape == ref.shape and float((ours - ref).abs().max()) < 2e-5


def test_xvector_tdnn_and_stats_pooling_equal_torch_nn_modules():
    """oracle/pyannote_ref.py::xvector_forward (functional conv1d / batch_norm / hand-written weighted statistics) vs nn.Conv1d with
    dilation + nn.LeakyReLU + nn.BatchNorm1d (eval) modules and torch.mean / torch.std pooling; the weighted pooling against a direct
    numpy evaluation of pyannote StatsPool's formulas."""
    from clearconverse_amd.weights import synthetic_xvector_state_dict
    sd = synthetic_xvector_state_dict(seed=12)
    wav = torch.from_numpy(synthetic_clip(8, 10.0)[:32000].copy())[None]
    with torch.no_grad():
        x = P.sincnet_forward(sd, wav[None])
        cin = 60
        for i, (cout, k, dil) in enumerate(zip((512, 512, 512, 512, 1500), (5, 3, 3, 1, 1), (1, 2, 3, 1, 1))):
            conv = torch.nn.Conv1d(cin, cout, k, dilation=dil)
            conv.load_state_dict({"weight": sd[f"tdnns.{i}.0.weight"], "bias": sd[f"tdnns.{i}.0.bias"]})
            bn = torch.nn.BatchNorm1d(cout).eval()
            bn.load_state_dict({kk: sd[f"tdnns.{i}.2.{kk}"] for kk in ("weight", "bias", "running_mean", "running_var")}, strict=False)
            x = bn(torch.nn.LeakyReLU()(conv(x)))
            cin = cout
        pooled = torch.cat([x.mean(dim=-1), x.std(dim=-1, unbiased=True)], dim=-1)
        ref = torch.nn.functional.linear(pooled, sd["embedding.weight"], sd["embedding.bias"])[0]
        ours = P.xvector_forward(sd, wav)
        assert float((ours - ref).norm() / ref.norm()) < 1e-5
        # weighted pooling: numpy evaluation of mean = sum(w x) / (sum w + eps), var = sum(w (x - mean)^2) / (v1 - v2 / v1 + eps)
        w = (torch.rand(x.shape[-1], generator=torch.Generator().manual_seed(1)) > 0.4).float()
        ours_w = P.xvector_forward(sd, wav, weights=w)
        xx, ww = x[0].double().numpy(), w.double().numpy()
        v1 = ww.sum() + 1e-8
        mean = (xx * ww).sum(1) / v1
        var = (((xx - mean[:, None]) ** 2) * ww).sum(1) / (v1 - (ww ** 2).sum() / v1 + 1e-8)
        ref_w = np.concatenate([mean, np.sqrt(var)]) @ sd["embedding.weight"].double().numpy().T + sd["embedding.bias"].double().numpy()
        assert float(np.linalg.norm(ours_w.double().numpy() - ref_w) / np.linalg.norm(ref_w)) < 1e-5


def test_wespeaker_resnet34_trunk_equals_a_network_of_torch_nn_modules():
    """oracle/wespeaker_ref.py::resnet_trunk (functional conv2d / batch_norm over state-dict keys) vs a ResNet-34 assembled from
    nn.Conv2d / nn.BatchNorm2d / nn.ReLU MODULES (BasicBlock: conv3x3-bn-relu-conv3x3-bn + shortcut, stages of 3-4-6-3 blocks with
    32-64-128-256 channels, stride 2 from stage 2 on -- the published wespeaker topology) loaded with the same tensors, and TSTP
    pooling vs torch.mean / torch.std."""
    from clearconverse_amd.weights import synthetic_resnet34_state_dict
    from oracle import wespeaker_ref as W
    sd = synthetic_resnet34_state_dict(seed=4)

    class Block(torch.nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(cin, cout, 3, stride, 1, bias=False); self.bn1 = torch.nn.BatchNorm2d(cout)
            self.conv2 = torch.nn.Conv2d(cout, cout, 3, 1, 1, bias=False); self.bn2 = torch.nn.BatchNorm2d(cout)
            self.shortcut = torch.nn.Sequential()
            if stride != 1 or cin != cout:
                self.shortcut = torch.nn.Sequential(torch.nn.Conv2d(cin, cout, 1, stride, bias=False), torch.nn.BatchNorm2d(cout))

        def forward(self, x):
            out = torch.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return torch.relu(out + self.shortcut(x))

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(1, 32, 3, 1, 1, bias=False); self.bn1 = torch.nn.BatchNorm2d(32)
            cin = 32
            for li, (n, c) in enumerate(zip((3, 4, 6, 3), (32, 64, 128, 256)), start=1):
                blocks = []
                for bi in range(n):
                    blocks.append(Block(cin, c, 2 if (li > 1 and bi == 0) else 1)); cin = c
                setattr(self, f"layer{li}", torch.nn.Sequential(*blocks))

        def forward(self, x):
            x = torch.relu(self.bn1(self.conv1(x)))
            for li in range(1, 5):
                x = getattr(self, f"layer{li}")(x)
            return x

    net = Net().eval()
    own = {k[len("resnet."):]: v for k, v in sd.items() if k.startswith("resnet.") and not k.startswith("resnet.seg_1")}
    missing, unexpected = net.load_state_dict(own, strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (missing, unexpected)
    wave = synthetic_clip(9, 10.0)[:48000]
    feats = torch.from_numpy(W.compute_fbank(wave))[None]
    with torch.no_grad():
        ours = W.resnet_trunk(sd, feats)
        y = net(feats.permute(0, 2, 1).unsqueeze(1))
        ref = y.reshape(y.shape[0], -1, y.shape[-1])
        assert ours.shape == ref.shape and float((ours - ref).norm() / ref.norm()) < 1e-5
        pooled = W.stats_pool(ours, None)
        assert torch.allclose(pooled, torch.cat([ref.mean(-1), ref.std(-1, unbiased=True)], dim=-1), rtol=1e-5, atol=1e-6)


def test_sepformer_encoder_decoder_and_positional_encoding_equal_independent_forms():
    """oracle/sepformer_ref.py: the learned encoder / decoder (functional conv1d / conv_transpose1d) vs nn.Conv1d / nn.ConvTranspose1d
    modules with the same weights, and the positional encoding vs the interleaved sin / cos table of "Attention is all you need"
    evaluated directly (SpeechBrain's PositionalEncoding)."""
    dims = SepDims(n_layers=1, n_blocks=1)
    sd = synthetic_sepformer_state_dict(dims, seed=2)
    x = torch.from_numpy(synthetic_clip(3, 10.0)[:8000].copy())[None]
    enc = torch.nn.Conv1d(1, dims.n_filters, dims.kernel, stride=dims.stride, bias=False)
    enc.load_state_dict({"weight": sd["encoder.conv1d.weight"]})
    dec = torch.nn.ConvTranspose1d(dims.n_filters, 1, dims.kernel, stride=dims.stride, bias=False)
    dec.load_state_dict({"weight": sd["decoder.weight"]})
    with torch.no_grad():
        w = torch.relu(enc(x[:, None]))
        a = F.relu(F.conv1d(x[:, None], sd["encoder.conv1d.weight"], None, stride=dims.stride))
        assert torch.equal(w, a)
        assert torch.allclose(dec(w)[0, 0], F.conv_transpose1d(a, sd["decoder.weight"], None, stride=dims.stride)[0, 0], atol=1e-6)
    pe = S.positional_encoding(150, 128).double().numpy()
    pos = np.arange(150)[:, None]
    div = np.exp(np.arange(0, 128, 2) * -(math.log(10000.0) / 128))
    want = np.zeros((150, 128)); want[:, 0::2] = np.sin(pos * div); want[:, 1::2] = np.cos(pos * div)
    assert np.abs(pe.reshape(150, 128) - want).max() < 2e-5          # the oracle evaluates position x frequency in fp32 (as upstream)


def test_centroid_linkage_dendrogram_equals_a_naive_agglomeration():
    """The clustering of the diarization pipeline leans on scipy's `linkage(method="centroid")` + `fcluster(criterion="distance")` in
    the oracle AND the product (as upstream does): checked here against a naive O(n^3) agglomeration written from the definition --
    repeatedly merge the two clusters whose CENTROIDS are closest (euclidean), stop when the closest pair is farther than the
    threshold -- on well-separated unit-normalised blobs (centroid linkage is not monotone, so the comparison uses blob data where
    the cut is unambiguous)."""
    from scipy.cluster.hierarchy import fcluster, linkage
    rng = np.random.default_rng(7)
    for trial in range(10):
        k = int(rng.integers(2, 5))
        cents = rng.standard_normal((k, 16)) * 3
        x = np.concatenate([c + 0.1 * rng.standard_normal((int(rng.integers(3, 9)), 16)) for c in cents])
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        thr = 0.7045654963945799
        lab = fcluster(linkage(x, method="centroid", metric="euclidean"), thr, criterion="distance")
        groups = [[i] for i in range(len(x))]
        while len(groups) > 1:
            cs = [x[g].mean(0) for g in groups]
            best, pair = None, None
            for i in range(len(groups)):
                for j in range(i + 1, len(groups)):
                    d = float(np.linalg.norm(cs[i] - cs[j]))
                    if best is None or d < best:
                        best, pair = d, (i, j)
            if best > thr:
                break
            i, j = pair
            groups[i] = groups[i] + groups[j]
            del groups[j]
        mine = np.zeros(len(x), dtype=int)
        for gi, g_ in enumerate(groups):
            mine[g_] = gi
        # same partition up to label names
        assert len(set(lab)) == len(groups) == k, trial
        for g_ in groups:
            assert len(set(lab[g_])) == 1, trial
