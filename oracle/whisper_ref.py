"""CPU restatement (plain PyTorch fp32) of the Whisper path the reference drives.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg as the checker; never by the product path (clearconverse_amd/).

What it restates: `whisper_model.transcribe(...)` as the reference calls it
(/root/reference/back/api.py:1286-1292, 1432-1438, 1474-1480; model loaded at 665-703).  The
arithmetic lives in the third-party package `openai-whisper`, which the reference imports
un-pinned (/root/reference/back/requirements.txt:12-19) and which is NOT vendored under
/root/reference and NOT installed here.  The functions below therefore restate the published
algorithm of openai-whisper (audio.py, model.py, decoding.py, transcribe.py) from recollection
[UPSTREAM-RECALL]; each function names the upstream function it follows.

PARITY STATUS: **parity unpinned** against the reference's own dependency (the reference holds no
tests, golden vectors or fixtures for this path: SURVEY.md section 4).  What pins this oracle
instead: tests/test_oracle_whisper.py cross-checks it against the independently written
`transformers` (5.15, installed locally) Whisper implementation with shared seeded weights
(log-mel vs WhisperFeatureExtractor, encoder/decoder logits vs WhisperForConditionalGeneration).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
N_SAMPLES = 480000
N_FRAMES = 3000


@dataclass
class Dims:
    n_mels: int = 80
    n_audio_ctx: int = 1500
    n_audio_state: int = 768
    n_audio_head: int = 12
    n_audio_layer: int = 12
    n_vocab: int = 51864
    n_text_ctx: int = 448
    n_text_state: int = 768
    n_text_head: int = 12
    n_text_layer: int = 12


@dataclass
class Rules:
    """Token ids of the English-only tokenizer (openai-whisper tokenizer.py) [UPSTREAM-RECALL]."""
    eot: int = 50256
    sot: int = 50257
    sot_prev: int = 50360
    no_speech: int = 50361
    no_timestamps: int = 50362
    timestamp_begin: int = 50363
    blank: int = 220  # GPT-2 id of " "
    max_initial_timestamp_index: int = 50
    suppress: Sequence[int] = ()


# ----------------------------------------------------------------------------------------------
# audio.py
# ----------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    mel = 3.0 * f / 200.0
    log_t = f >= 1000.0
    mel = np.where(log_t, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), mel)
    return mel


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f = 200.0 * m / 3.0
    log_t = m >= 15.0
    return np.where(log_t, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)


def mel_filters(n_mels: int = 80, n_fft: int = N_FFT, sr: int = SAMPLE_RATE) -> np.ndarray:
    """librosa.filters.mel(sr=16000, n_fft=400, n_mels=80) (slaney scale, slaney norm) -- the
    contents of openai-whisper's assets/mel_filters.npz, which audio.py::mel_filters loads."""
    fft_freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_pts = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_pts)
    ramps = mel_pts[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_pts[2:n_mels + 2] - mel_pts[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def log_mel_spectrogram(audio: torch.Tensor, n_mels: int = 80, padding: int = N_SAMPLES) -> torch.Tensor:
    """audio.py::log_mel_spectrogram -- audio [n] f32 -> [n_mels, (n+padding)//160] f32."""
    audio = audio.to(torch.float32)
    if padding > 0:
        audio = F.pad(audio, (0, padding))
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    filters = torch.from_numpy(mel_filters(n_mels))
    mel_spec = filters @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec


def pad_or_trim(x: torch.Tensor, length: int = N_FRAMES) -> torch.Tensor:
    """audio.py::pad_or_trim along the last axis (zero padding)."""
    if x.shape[-1] > length:
        return x[..., :length]
    if x.shape[-1] < length:
        return F.pad(x, (0, length - x.shape[-1]))
    return x


# ----------------------------------------------------------------------------------------------
# model.py
# ----------------------------------------------------------------------------------------------
def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """model.py::sinusoids -- first half sin, second half cos."""
    log_inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-log_inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


class WhisperRef:
    """Functional restatement of model.py::Whisper over an openai-layout state dict."""

    def __init__(self, dims: Dims, sd: Dict[str, torch.Tensor]):
        self.dims = dims
        self.sd = {k: v.detach().to(torch.float32) for k, v in sd.items()}

    # -- helpers --
    def _lin(self, x, name, bias=True):
        return F.linear(x, self.sd[name + ".weight"], self.sd[name + ".bias"] if bias else None)

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[-1],), self.sd[name + ".weight"], self.sd[name + ".bias"], 1e-5)

    def _qkv_attention(self, q, k, v, n_head, mask=None):
        """model.py::MultiHeadAttention.qkv_attention: scale d_head^-0.25 on q and on k."""
        B, T, D = q.shape
        scale = (D // n_head) ** -0.25
        q = q.view(B, T, n_head, -1).permute(0, 2, 1, 3) * scale
        k = k.view(B, k.shape[1], n_head, -1).permute(0, 2, 3, 1) * scale
        v = v.view(B, v.shape[1], n_head, -1).permute(0, 2, 1, 3)
        qk = q @ k
        if mask is not None:
            qk = qk + mask[:T, :T]
        w = F.softmax(qk.float(), dim=-1)
        return (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)

    def _attn(self, x, prefix, n_head, xa=None, mask=None):
        q = self._lin(x, prefix + ".query")
        src = x if xa is None else xa
        k = self._lin(src, prefix + ".key", bias=False)
        v = self._lin(src, prefix + ".value")
        return self._lin(self._qkv_attention(q, k, v, n_head, mask), prefix + ".out")

    def _mlp(self, x, prefix):
        return self._lin(F.gelu(self._lin(x, prefix + ".mlp.0")), prefix + ".mlp.2")

    # -- AudioEncoder.forward --
    def encode(self, mel: torch.Tensor, return_layers: bool = False):
        """mel [B, n_mels, 3000] -> [B, 1500, D]."""
        d = self.dims
        x = F.gelu(F.conv1d(mel, self.sd["encoder.conv1.weight"], self.sd["encoder.conv1.bias"], padding=1))
        x = F.gelu(F.conv1d(x, self.sd["encoder.conv2.weight"], self.sd["encoder.conv2.bias"], stride=2, padding=1))
        x = x.permute(0, 2, 1)
        x = x + self.sd["encoder.positional_embedding"]
        layers = [x]
        for l in range(d.n_audio_layer):
            p = f"encoder.blocks.{l}"
            x = x + self._attn(self._ln(x, p + ".attn_ln"), p + ".attn", d.n_audio_head)
            x = x + self._mlp(self._ln(x, p + ".mlp_ln"), p)
            layers.append(x)
        x = self._ln(x, "encoder.ln_post")
        return (x, layers) if return_layers else x

    # -- TextDecoder.forward (no kv cache: full recompute, fine for an oracle) --
    def decoder_logits(self, tokens: torch.Tensor, xa: torch.Tensor) -> torch.Tensor:
        """tokens [B, T] int64, xa [B, 1500, D] -> logits [B, T, V] f32."""
        d = self.dims
        T = tokens.shape[-1]
        x = self.sd["decoder.token_embedding.weight"][tokens] + self.sd["decoder.positional_embedding"][:T]
        mask = torch.full((d.n_text_ctx, d.n_text_ctx), float("-inf")).triu_(1)
        for l in range(d.n_text_layer):
            p = f"decoder.blocks.{l}"
            x = x + self._attn(self._ln(x, p + ".attn_ln"), p + ".attn", d.n_text_head, mask=mask)
            x = x + self._attn(self._ln(x, p + ".cross_attn_ln"), p + ".cross_attn", d.n_text_head, xa=xa)
            x = x + self._mlp(self._ln(x, p + ".mlp_ln"), p)
        x = self._ln(x, "decoder.ln")
        return (x @ self.sd["decoder.token_embedding.weight"].T).float()


# ----------------------------------------------------------------------------------------------
# decoding.py
# ----------------------------------------------------------------------------------------------
def apply_filters(logits: torch.Tensor, sampled: List[int], rules: Rules) -> torch.Tensor:
    """SuppressBlank, SuppressTokens, ApplyTimestampRules in that order (decoding.py) for ONE row.
    `sampled` = tokens sampled so far (tokens[sample_begin:])."""
    logits = logits.clone()
    ninf = float("-inf")
    tsb = rules.timestamp_begin
    if len(sampled) == 0:  # SuppressBlank
        logits[rules.blank] = ninf
        logits[rules.eot] = ninf
    if len(rules.suppress):  # SuppressTokens
        logits[torch.as_tensor(list(rules.suppress), dtype=torch.long)] = ninf
    # ApplyTimestampRules
    logits[rules.no_timestamps] = ninf
    last_was_ts = len(sampled) >= 1 and sampled[-1] >= tsb
    pen_was_ts = len(sampled) < 2 or sampled[-2] >= tsb
    if last_was_ts:
        if pen_was_ts:
            logits[tsb:] = ninf
        else:
            logits[: rules.eot] = ninf
    ts = [t for t in sampled if t >= tsb]
    if ts:
        if last_was_ts and not pen_was_ts:
            ts_last = ts[-1]
        else:
            ts_last = ts[-1] + 1
        logits[tsb:ts_last] = ninf
    if len(sampled) == 0:
        logits[:tsb] = ninf
        if rules.max_initial_timestamp_index is not None and rules.max_initial_timestamp_index >= 0:
            logits[tsb + rules.max_initial_timestamp_index + 1:] = ninf
    logprobs = F.log_softmax(logits.float(), dim=-1)
    ts_logprob = logprobs[tsb:].logsumexp(dim=-1)
    max_text = logprobs[:tsb].max()
    if ts_logprob > max_text:
        logits[:tsb] = ninf
    return logits


@dataclass
class DecodeResult:
    tokens: List[int]
    sum_logprob: float
    avg_logprob: float
    no_speech_prob: float
    # per-step diagnostics for tolerance-aware parity: filtered logits margins (top1 - top2)
    margins: List[float]


def greedy_decode(model: WhisperRef, xa: torch.Tensor, prompts: List[List[int]], rules: Rules,
                  sample_len: int = 224, forced: Optional[List[List[int]]] = None) -> List[DecodeResult]:
    """decoding.py::DecodingTask._main_loop with GreedyDecoder (temperature 0), one row at a time.

    prompts[b] = initial tokens (sot_prev + prompt + sot).  If `forced` is given, row b follows
    forced[b] instead of its own argmax (teacher forcing) while margins/logprobs are still taken
    from this model -- used for tolerance-aware parity against the bf16 GPU path."""
    out = []
    for b, prompt in enumerate(prompts):
        toks = list(prompt)
        sampled: List[int] = []
        sum_lp = 0.0
        no_speech = 0.0
        margins = []
        xab = xa[b:b + 1]
        n_steps = sample_len if forced is None else min(sample_len, len(forced[b]) + 1)
        for i in range(n_steps):
            logits = model.decoder_logits(torch.tensor([toks], dtype=torch.long), xab)[0]
            if i == 0:
                no_speech = F.softmax(logits[len(prompt) - 1].float(), dim=-1)[rules.no_speech].item()
            lg = apply_filters(logits[-1], sampled, rules)
            lp = F.log_softmax(lg.float(), dim=-1)
            top2 = torch.topk(lg, 2).values
            margins.append(float(top2[0] - top2[1]))
            nxt = int(lg.argmax())
            if forced is not None:
                if i >= len(forced[b]):
                    break
                nxt = forced[b][i]
            sum_lp += float(lp[nxt])
            sampled.append(nxt)
            toks.append(nxt)
            if nxt == rules.eot or len(toks) > model.dims.n_text_ctx:
                break
        text = sampled[: sampled.index(rules.eot)] if rules.eot in sampled else sampled
        out.append(DecodeResult(text, sum_lp, sum_lp / (len(text) + 1), no_speech, margins))
    return out


# ----------------------------------------------------------------------------------------------
# KV-cached incremental decoding (what openai-whisper's PyTorchInference does with its hooks):
# same arithmetic as decoder_logits, one token per call.  Used for the timed CPU baseline and for
# long greedy runs in tests.
# ----------------------------------------------------------------------------------------------
class CachedDecoder:
    def __init__(self, model: WhisperRef, xa: torch.Tensor):
        self.m = model
        d = model.dims
        self.cross = []
        for l in range(d.n_text_layer):
            p = f"decoder.blocks.{l}.cross_attn"
            self.cross.append((model._lin(xa, p + ".key", bias=False), model._lin(xa, p + ".value")))
        self.self_k = [None] * d.n_text_layer
        self.self_v = [None] * d.n_text_layer
        self.offset = 0

    def step(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens [B, t_new] -> logits [B, t_new, V]; appends to the self-attention cache."""
        m, d = self.m, self.m.dims
        t_new = tokens.shape[-1]
        x = m.sd["decoder.token_embedding.weight"][tokens] + m.sd["decoder.positional_embedding"][self.offset:self.offset + t_new]
        T = self.offset + t_new
        mask = torch.full((t_new, T), float("-inf")).triu_(self.offset + 1)
        for l in range(d.n_text_layer):
            p = f"decoder.blocks.{l}"
            h = m._ln(x, p + ".attn_ln")
            q = m._lin(h, p + ".attn.query")
            k = m._lin(h, p + ".attn.key", bias=False)
            v = m._lin(h, p + ".attn.value")
            self.self_k[l] = k if self.self_k[l] is None else torch.cat([self.self_k[l], k], dim=1)
            self.self_v[l] = v if self.self_v[l] is None else torch.cat([self.self_v[l], v], dim=1)
            B, _, D = q.shape
            nh = d.n_text_head
            sc = (D // nh) ** -0.25
            qh = q.view(B, t_new, nh, -1).permute(0, 2, 1, 3) * sc
            kh = self.self_k[l].view(B, T, nh, -1).permute(0, 2, 3, 1) * sc
            vh = self.self_v[l].view(B, T, nh, -1).permute(0, 2, 1, 3)
            w = F.softmax((qh @ kh + mask).float(), dim=-1)
            x = x + m._lin((w @ vh).permute(0, 2, 1, 3).flatten(start_dim=2), p + ".attn.out")
            h = m._ln(x, p + ".cross_attn_ln")
            q = m._lin(h, p + ".cross_attn.query")
            ck, cv = self.cross[l]
            x = x + m._lin(m._qkv_attention(q, ck, cv, nh), p + ".cross_attn.out")
            x = x + m._mlp(m._ln(x, p + ".mlp_ln"), p)
        self.offset = T
        x = m._ln(x, "decoder.ln")
        return (x @ m.sd["decoder.token_embedding.weight"].T).float()


# ---------------------------------------------------------------------------------------------
# temperature > 0 (decoding.py::GreedyDecoder.update: Categorical(logits=logits / temperature).sample()).
# torch's sampler stream cannot be reproduced on a GPU kernel, so both sides draw by the Gumbel-max trick
# over the SAME counter-based noise: Philox4x32-10 keyed by (seed, row, step, token id).
# ---------------------------------------------------------------------------------------------
def philox4x32(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """Philox4x32-10 (Salmon et al., SC'11).  ctr [..., 4] uint32, key [2] uint32 -> [..., 4] uint32."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask, p0 & mask]
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def gumbel_noise(seed: int, row: int, step: int, n_vocab: int) -> np.ndarray:
    """Standard Gumbel noise g[v] for every vocabulary id of one (row, step): the draw for id v is word v % 4 of
    philox(counter = (v // 4, row, step, 0), key = (seed lo, seed hi)); u = (top 24 bits + 0.5) / 2^24."""
    nq = (n_vocab + 3) // 4
    ctr = np.zeros((nq, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nq, dtype=np.uint32)
    ctr[:, 1] = row
    ctr[:, 2] = step
    x = philox4x32(ctr, np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)).reshape(-1)[:n_vocab]
    u = ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    return (-np.log(-np.log(u))).astype(np.float32)


def sample_token(filtered_logits: torch.Tensor, temperature: float, seed: int, row: int, step: int):
    """One GreedyDecoder.update step on already filtered logits [V]: returns (token, log-prob under the UNSCALED
    log-softmax, Gumbel-perturbed scores) -- argmax of logits / T + g is a Categorical(logits / T) sample."""
    lg = filtered_logits.float()
    lp = F.log_softmax(lg, dim=-1)
    if temperature == 0:
        nxt = int(lg.argmax())
        return nxt, float(lp[nxt]), lg
    score = lg / temperature + torch.from_numpy(gumbel_noise(seed, row, step, lg.shape[-1]))
    nxt = int(score.argmax())
    return nxt, float(lp[nxt]), score


def greedy_decode_cached(model: WhisperRef, xa: torch.Tensor, prompt: List[int], rules: Rules,
                         sample_len: int = 224) -> DecodeResult:
    """Greedy decode of ONE sequence with the KV cache (identical filters / bookkeeping to greedy_decode)."""
    dec = CachedDecoder(model, xa)
    logits = dec.step(torch.tensor([prompt], dtype=torch.long))[0]
    no_speech = F.softmax(logits[len(prompt) - 1].float(), dim=-1)[rules.no_speech].item()
    sampled: List[int] = []
    sum_lp = 0.0
    margins = []
    last = logits[-1]
    n_tok = len(prompt)
    for i in range(sample_len):
        lg = apply_filters(last, sampled, rules)
        lp = F.log_softmax(lg.float(), dim=-1)
        top2 = torch.topk(lg, 2).values
        margins.append(float(top2[0] - top2[1]))
        nxt = int(lg.argmax())
        sum_lp += float(lp[nxt])
        sampled.append(nxt)
        n_tok += 1
        if nxt == rules.eot or n_tok > model.dims.n_text_ctx or i + 1 == sample_len:
            break
        last = dec.step(torch.tensor([[nxt]], dtype=torch.long))[0, -1]
    text = sampled[: sampled.index(rules.eot)] if rules.eot in sampled else sampled
    return DecodeResult(text, sum_lp, sum_lp / (len(text) + 1), no_speech, margins)
