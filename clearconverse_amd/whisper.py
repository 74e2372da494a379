"""`WhisperModel`: drop-in for the `self.whisper_model` object of the reference
(created at /root/reference/back/api.py:665-703; called at 1286-1292, 1432-1438, 1474-1480).

Only `.transcribe(audio_np, initial_prompt=, word_timestamps=, condition_on_previous_text=,
temperature=)['text']` is consumed by the reference (back/api.py:1103, 1447, 1488); the same call
shape is kept.  All arithmetic runs in libccx (hand-written HIP for gfx950) through the C ABI; this
module holds only the window/segment bookkeeping of openai-whisper's transcribe.py
[UPSTREAM-RECALL] and a batch entry point the reference lacks.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import asdict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .audio import mel_filterbank
from .tokenizer import DecodeRules, IdTokenizer, get_tokenizer
from .weights import WhisperDims

N_FRAMES = 3000
HOP = 160
SAMPLE_RATE = 16000
N_SAMPLES = 480000
FRAMES_PER_SECOND = 100
TIME_PRECISION = 0.02  # seconds per timestamp token
INPUT_STRIDE = 2       # mel frames per encoder position
CROSS_PATHS = {0: "kv16", 1: "kv_stream", 2: "xa_stream"}


class WindowLoop:
    """Host half of openai-whisper's `transcribe()` for ONE clip [UPSTREAM-RECALL: whisper/transcribe.py, main loop]: which 30 s window
    is decoded next and with which prompt, and what a decoded window adds to the segments / tokens / `text` -- the only key the
    reference reads (back/api.py:1103, 1447, 1488).  No GPU state: `WhisperModel.transcribe_batch` owns the device work and calls
    `next_window()` / `advance()`; tests/test_transcribe_loop_cpu.py drives the same object with scripted decode results against
    oracle/whisper_transcribe_ref.py.

    Deviation: with `word_timestamps=True` (back/api.py:1435, 1477) upstream aligns words by cross-attention DTW
    and, when a window does not end on a single timestamp, moves `seek` to the end of the last aligned word instead of the last
    timestamp token.  The DTW (K11) is not built -- its words are never read by the reference -- so `seek` keeps the timestamp-token
    rule; only audio that needs more than one window (longer than 30 s, or a window that ends inside an unfinished segment) can see it."""

    def __init__(self, rules: DecodeRules, tokenizer, content_frames: int, initial_prompt: Optional[str], n_text_ctx: int,
                 condition_on_previous_text: bool = True, no_speech_threshold: Optional[float] = 0.6,
                 logprob_threshold: Optional[float] = -1.0):
        self.rules, self.tokenizer, self.n_text_ctx = rules, tokenizer, int(n_text_ctx)
        self.content = int(content_frames)
        self.condition, self.no_speech_threshold, self.logprob_threshold = condition_on_previous_text, no_speech_threshold, logprob_threshold
        ipt = tokenizer.encode(" " + initial_prompt.strip()) if initial_prompt else []
        self.seek, self.all_tokens, self.n_init, self.reset = 0, list(ipt), len(ipt), 0
        self.segments: List[dict] = []
        self.seeks: List[int] = []

    def active(self) -> bool:
        return self.seek < self.content

    def prompt_tokens(self) -> List[int]:
        """decode_options["prompt"] = all_tokens[prompt_reset_since:]"""
        return self.all_tokens[self.reset:]

    def initial_tokens(self) -> List[int]:
        """decoding.py::_get_initial_tokens: [sot_prev] + prompt[-(n_ctx // 2 - 1):] + [sot]."""
        p = self.prompt_tokens()
        toks: List[int] = []
        if len(p):
            toks = [self.rules.sot_prev] + list(p)[-(self.n_text_ctx // 2 - 1):]
        return toks + [self.rules.sot]

    def advance(self, r: dict, temperature: float = 0.0) -> None:
        """One decoded window (r: tokens before eot, avg_logprob, no_speech_prob) -> segments, tokens, the next seek."""
        tsb, eot = self.rules.timestamp_begin, self.rules.eot
        seek = self.seek
        self.seeks.append(seek)
        segment_size = min(N_FRAMES, self.content - seek)
        time_offset = seek * HOP / SAMPLE_RATE
        segment_duration = segment_size * HOP / SAMPLE_RATE
        tokens = list(r["tokens"])
        if self.no_speech_threshold is not None:
            skip = r["no_speech_prob"] > self.no_speech_threshold
            if self.logprob_threshold is not None and r["avg_logprob"] > self.logprob_threshold:
                skip = False
            if skip:
                self.seek = seek + segment_size
                return
        is_ts = [t >= tsb for t in tokens]
        single_ts_ending = is_ts[-2:] == [False, True]
        consecutive = [i + 1 for i in range(len(tokens) - 1) if is_ts[i] and is_ts[i + 1]]
        new_segments = []

        def add(start, end, toks):
            new_segments.append(dict(seek=seek, start=start, end=end, tokens=list(toks),
                                     text=self.tokenizer.decode([t for t in toks if t < eot])))

        if consecutive:
            slices = list(consecutive)
            if single_ts_ending:
                slices.append(len(tokens))
            last = 0
            for cur in slices:
                sl = tokens[last:cur]
                add(time_offset + (sl[0] - tsb) * TIME_PRECISION, time_offset + (sl[-1] - tsb) * TIME_PRECISION, sl)
                last = cur
            if single_ts_ending:
                self.seek = seek + segment_size
            else:
                self.seek = seek + (tokens[last - 1] - tsb) * INPUT_STRIDE
        else:
            duration = segment_duration
            ts = [t for t in tokens if t >= tsb]
            if ts and ts[-1] != tsb:
                duration = (ts[-1] - tsb) * TIME_PRECISION
            add(time_offset, time_offset + duration, tokens)
            self.seek = seek + segment_size
        for s in new_segments:
            # "if a segment is instantaneous or does not contain text, clear it": its tokens do not reach the prompt or the text
            if s["start"] == s["end"] or s["text"].strip() == "":
                s["text"], s["tokens"] = "", []
            self.segments.append(s)
            self.all_tokens.extend(s["tokens"])
        if not self.condition or temperature > 0.5:      # "do not feed the prompt tokens if a high temperature was used"
            self.reset = len(self.all_tokens)
        if self.seek <= seek:  # cannot happen under ApplyTimestampRules (a closing timestamp is > its opening one); never loop forever
            self.seek = seek + segment_size

    def result(self) -> dict:
        text_tokens = self.all_tokens[self.n_init:]
        return dict(text=self.tokenizer.decode(text_tokens), segments=self.segments, language="en", tokens=list(text_tokens))


class WhisperModel:
    def __init__(self, dims: WhisperDims, state_dict: Dict[str, torch.Tensor], max_batch: int = 8,
                 device: int = 0, rules: Optional[DecodeRules] = None, tokenizer=None,
                 ctx: Optional[_lib.Context] = None, max_audio_seconds: float = 30.0,
                 share_encoder_scratch_with: Optional["WhisperModel"] = None):
        if not torch.cuda.is_available():
            raise _lib.CcxError("WhisperModel needs a ROCm GPU: the HIP path has no CPU fallback")
        self.dims = dims
        self.max_batch = int(max_batch)
        self.device = torch.device("cuda", device)
        self.ctx = ctx or _lib.Context(device)
        self.lib = self.ctx.lib
        self.rules = rules or DecodeRules()
        self.tokenizer = tokenizer or get_tokenizer()
        h = C.c_void_p()
        cd = _lib.WhisperDims(**asdict(dims))
        self.ctx.check(self.lib.ccx_whisper_create(self.ctx.handle, C.byref(cd), self.max_batch, C.byref(h)),
                       "ccx_whisper_create")
        self.handle = h
        self.max_audio_seconds = float(max_audio_seconds)
        self.sample_seed, self._sample_calls = 0, 0     # temperature > 0: Philox seed and per-call counter
        self.last_cross_path = None
        self.ctx.check(self.lib.ccx_whisper_set_max_audio(self.handle, self.max_audio_seconds), "ccx_whisper_set_max_audio")
        # log-mel / encoder workspaces of another instance (kept alive here): only for instances whose log_mel / encode calls
        # are ordered on one stream, as in BatchPipeline.run_pinned_pipelined (include/ccx.h)
        self._scratch_donor = share_encoder_scratch_with
        if share_encoder_scratch_with is not None:
            self.ctx.check(self.lib.ccx_whisper_share_encoder_scratch(self.handle, share_encoder_scratch_with.handle),
                           "ccx_whisper_share_encoder_scratch")
        self._load(state_dict)
        self.set_rules(self.rules)

    # ------------------------------------------------------------------ weights / rules
    def _load(self, sd: Dict[str, torch.Tensor]):
        tensors = dict(sd)
        tensors["mel_filters"] = torch.from_numpy(mel_filterbank(self.dims.n_mels))
        for name, t in tensors.items():
            t = t.detach().to("cpu")
            if t.dtype == torch.float32:
                code = 0
            elif t.dtype == torch.bfloat16:
                code = 1
            elif t.dtype == torch.float16:
                code = 2
            else:
                t, code = t.float(), 0
            t = t.contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            self.ctx.check(self.lib.ccx_whisper_set_tensor(self.handle, name.encode(), t.data_ptr(), code, t.dim(), shape),
                           f"set_tensor({name})")
        self.ctx.check(self.lib.ccx_whisper_finalize(self.handle), "ccx_whisper_finalize")

    def set_rules(self, rules: DecodeRules):
        sup = (C.c_int * len(rules.suppress))(*[int(x) for x in rules.suppress])
        r = _lib.DecodeRules(rules.eot, rules.sot, rules.sot_prev, rules.no_speech, rules.no_timestamps,
                             rules.timestamp_begin, rules.blank, rules.max_initial_timestamp_index,
                             len(rules.suppress), sup)
        self.ctx.check(self.lib.ccx_whisper_set_rules(self.handle, C.byref(r)), "ccx_whisper_set_rules")
        self.rules = rules

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ccx_whisper_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ stage entry points
    def log_mel(self, audio: torch.Tensor, n_samples: Sequence[int], seek: Optional[Sequence[int]] = None,
                return_mel: bool = False) -> Optional[torch.Tensor]:
        """audio: [B, stride] f32 on the GPU.  Stages the conv-stem input inside the model."""
        B = audio.shape[0]
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.is_contiguous()
        ns = (C.c_int * B)(*[int(x) for x in n_samples])
        sk = (C.c_int * B)(*[int(x) for x in seek]) if seek is not None else None
        mel = torch.empty(B, self.dims.n_mels, N_FRAMES, device=self.device, dtype=torch.float32) if return_mel else None
        self.ctx.check(self.lib.ccx_whisper_logmel(self.handle, audio.data_ptr(), audio.shape[1], ns, sk, B,
                                                   _lib.ptr(mel), _lib.current_stream_ptr()), "ccx_whisper_logmel")
        return mel

    def set_mel(self, mel: torch.Tensor):
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous() and mel.shape[1:] == (self.dims.n_mels, N_FRAMES)
        self.ctx.check(self.lib.ccx_whisper_set_mel(self.handle, mel.data_ptr(), mel.shape[0], _lib.current_stream_ptr()),
                       "ccx_whisper_set_mel")

    def encode(self, B: int, return_xa: bool = False) -> Optional[torch.Tensor]:
        xa = torch.empty(B, self.dims.n_audio_ctx, self.dims.n_audio_state, device=self.device, dtype=torch.float32) if return_xa else None
        self.ctx.check(self.lib.ccx_whisper_encode(self.handle, B, _lib.ptr(xa), _lib.current_stream_ptr()), "ccx_whisper_encode")
        return xa

    def decoder_logits(self, tokens: np.ndarray) -> torch.Tensor:
        """Teacher-forced logits [B, T, V] for the currently encoded windows."""
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        B, T = tok.shape
        out = torch.empty(B, T, self.dims.n_vocab, device=self.device, dtype=torch.float32)
        self.ctx.check(self.lib.ccx_whisper_decoder_logits(self.handle, tok.ctypes.data_as(C.POINTER(C.c_int32)), B, T,
                                                           out.data_ptr(), _lib.current_stream_ptr()), "ccx_whisper_decoder_logits")
        return out

    def prepare_lanes(self, stream: Optional[torch.cuda.Stream] = None):
        """Pick the decode lanes' internal streams for decodes issued on `stream` (default: the current stream) now, on an
        idle device -- needed before decodes are overlapped with work on other streams (batch.run_pinned_pipelined)."""
        sp = int((stream or torch.cuda.current_stream()).cuda_stream)
        self.ctx.check(self.lib.ccx_whisper_prepare_lanes(self.handle, sp), "ccx_whisper_prepare_lanes")

    def trace_lanes(self, path: Optional[str], level: int = 1):
        """Switch the in-graph lane trace on (path) or off (None): ccx_whisper_trace_lanes.  Drops the captured step graphs."""
        self.ctx.check(self.lib.ccx_whisper_trace_lanes(self.handle, path.encode() if path else None, int(level)), "ccx_whisper_trace_lanes")

    def decode_greedy(self, prompts: Sequence[Sequence[int]], sample_len: Optional[int] = None) -> List[dict]:
        """Greedy DecodingTask.run over the currently encoded windows (temperature 0)."""
        return self.decode(prompts, sample_len, temperature=0.0)

    def decode(self, prompts: Sequence[Sequence[int]], sample_len: Optional[int] = None, temperature: float = 0.0,
               seed: int = 0) -> List[dict]:
        """DecodingTask.run over the currently encoded windows; prompts[b] are the full initial tokens
        (sot_prev + prompt + sot).  temperature 0: argmax.  temperature > 0: one Categorical(logits / T) sample per
        step (decoding.py::GreedyDecoder.update), drawn on the device from Philox noise keyed by
        (seed, row, step, token id) -- reproducible, not bit-equal to torch's sampler."""
        if not (temperature >= 0.0):
            raise _lib.CcxError("temperature must be >= 0")
        B = len(prompts)
        sample_len = sample_len or self.dims.n_text_ctx // 2
        mp = max(len(p) for p in prompts)
        ids = np.full((B, mp), self.rules.eot, dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        for b, p in enumerate(prompts):
            ids[b, :len(p)] = p
            lens[b] = len(p)
        toks = np.zeros((B, sample_len), dtype=np.int32)
        ntok = np.zeros(B, dtype=np.int32)
        slp = np.zeros(B, dtype=np.float32)
        nsp = np.zeros(B, dtype=np.float32)
        i32p, fp = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        self.ctx.check(self.lib.ccx_whisper_decode(
            self.handle, ids.ctypes.data_as(i32p), lens.ctypes.data_as(i32p), mp, B, sample_len, float(temperature),
            int(seed) & 0xFFFFFFFFFFFFFFFF, toks.ctypes.data_as(i32p), ntok.ctypes.data_as(i32p), slp.ctypes.data_as(fp),
            nsp.ctypes.data_as(fp), _lib.current_stream_ptr()), "ccx_whisper_decode")
        # which cross-attention formulation the decode ran (include/ccx.h: ccx_whisper_last_cross_path), kept in every record
        self.last_cross_path = CROSS_PATHS.get(int(self.lib.ccx_whisper_last_cross_path(self.handle)), "unknown")
        return [dict(tokens=toks[b, :ntok[b]].tolist(), sum_logprob=float(slp[b]),
                     avg_logprob=float(slp[b]) / (int(ntok[b]) + 1), no_speech_prob=float(nsp[b]),
                     cross_path=self.last_cross_path) for b in range(B)]

    # ------------------------------------------------------------------ transcribe (reference call surface)
    def initial_tokens(self, prompt_tokens: Sequence[int]) -> List[int]:
        """decoding.py::_get_initial_tokens: [sot_prev] + prompt[-(n_ctx//2 - 1):] + [sot]."""
        toks: List[int] = []
        if len(prompt_tokens):
            toks = [self.rules.sot_prev] + list(prompt_tokens)[-(self.dims.n_text_ctx // 2 - 1):]
        return toks + [self.rules.sot]

    def transcribe(self, audio, initial_prompt: Optional[str] = None, word_timestamps: bool = False,
                   condition_on_previous_text: bool = True, temperature: float = 0.0,
                   no_speech_threshold: Optional[float] = 0.6, logprob_threshold: Optional[float] = -1.0, **_ignored):
        """One clip, same signature as whisper.transcribe as the reference uses it.  temperature 0 is the
        parity mode (greedy); a positive float (the reference's Config.temperature = 0.1,
        back/api.py:128) samples every token from Categorical(logits / T) -- a single temperature means no
        fallback loop upstream either.  Draws are reproducible: seeded by `self.sample_seed` and a per-call
        counter.  word_timestamps only changes fields the reference never reads."""
        return self.transcribe_batch([audio], [initial_prompt], condition_on_previous_text=condition_on_previous_text,
                                     temperature=temperature, no_speech_threshold=no_speech_threshold,
                                     logprob_threshold=logprob_threshold)[0]

    def transcribe_batch(self, audios: Sequence, initial_prompts: Optional[Sequence[Optional[str]]] = None,
                         condition_on_previous_text: bool = True, temperature: float = 0.0,
                         no_speech_threshold: Optional[float] = 0.6, logprob_threshold: Optional[float] = -1.0) -> List[dict]:
        """Independent clips decoded together (each window of each clip is one sequence of a batch)."""
        if isinstance(temperature, (tuple, list)):
            raise _lib.CcxError("temperature fallback schedules are not implemented: pass one temperature (the reference does)")
        temperature = float(temperature)
        n = len(audios)
        initial_prompts = list(initial_prompts) if initial_prompts is not None else [None] * n
        # device tensors (the processor's crops) stay on the device; host arrays go up in one transfer
        on_dev = n > 0 and all(isinstance(a, torch.Tensor) and a.is_cuda for a in audios)
        if on_dev:
            clips = [a.detach().reshape(-1) for a in audios]
            lens = [int(c.numel()) for c in clips]
        else:
            clips = []
            for a in audios:
                a = a.detach().to("cpu").numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
                clips.append(np.ascontiguousarray(a.reshape(-1), dtype=np.float32))
            lens = [len(c) for c in clips]
        state = [WindowLoop(self.rules, self.tokenizer, lens[i] // HOP, initial_prompts[i], self.dims.n_text_ctx,
                            condition_on_previous_text, no_speech_threshold, logprob_threshold) for i in range(n)]
        stride = max(max(lens, default=1), 1)
        if stride > self.max_audio_seconds * SAMPLE_RATE:
            raise _lib.CcxError(f"clip of {stride / SAMPLE_RATE:.1f} s exceeds max_audio_seconds={self.max_audio_seconds}")
        # clips stay resident on the GPU for all windows
        if on_dev:
            dev_audio = torch.zeros(n, stride, device=self.device, dtype=torch.float32)
            for i, c in enumerate(clips):
                dev_audio[i, :lens[i]] = c.to(self.device, torch.float32)
        else:
            host = np.zeros((n, stride), dtype=np.float32)
            for i, c in enumerate(clips):
                host[i, :lens[i]] = c
            dev_audio = torch.from_numpy(host).to(self.device)
        while True:
            active = [i for i in range(n) if state[i].active()]
            if not active:
                break
            for c0 in range(0, len(active), self.max_batch):
                grp = active[c0:c0 + self.max_batch]
                if len(grp) == n:
                    a = dev_audio
                else:
                    a = dev_audio.index_select(0, torch.tensor(grp, device=self.device)).contiguous()
                self.log_mel(a, [lens[i] for i in grp], [state[i].seek for i in grp])
                self.encode(len(grp))
                prompts = [state[i].initial_tokens() for i in grp]
                self._sample_calls += 1
                results = self.decode(prompts, temperature=temperature, seed=(int(self.sample_seed) << 32) + self._sample_calls)
                for i, r in zip(grp, results):
                    state[i].advance(r, temperature)
        return [st.result() for st in state]
