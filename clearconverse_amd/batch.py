"""Clip-batched driver the reference lacks (`Config.transcription_batch_size` is
never read there; every task is one process, one clip, one call at a time).

`BatchPipeline.run_pinned` executes the complete op mix of `process_file` for B independent 30 s clips,
stage by stage, with every model call batched across clips.  Because seeded random weights give arbitrary
diarization / EOT, the control flow follows the pinned 30 s schedule of SURVEY.md section 8d (A 0-9 s,
B 7-16 s, A 18-24 s, B 26-30 s -> 2 overlap-bearing + 2 regular segments; VAD / diarization are COMPUTED
but do not steer): this is a build decision for benchmarking and is stated next to every number.
For real weights use `EnhancedAudioProcessor` (processor.py), which follows the models' own decisions.

Clips are independent units, so multi-GPU = clip i -> rank i mod world (one process per GPU), no
data-path collective; `gather_transcripts` is the one all-gather at the end (C2)."""
from __future__ import annotations

import ctypes as C
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .audio import SCHEDULE_30S
from .processor import PROMPT_SINGLE, PROMPT_TWO_PEOPLE

SR = 16000


class BatchPipeline:
    def __init__(self, models: Dict[str, object], sliding_window: float = 0.8, sliding_step: float = 0.4,
                 noise_reduction_amount: float = 0.5, whisper_group: int = 64, sample_len: int = 224):
        self.m = models
        self.ctx: _lib.Context = models["ctx"]
        self.win, self.hop, self.nra = sliding_window, sliding_step, noise_reduction_amount
        self.group, self.sample_len = whisper_group, sample_len
        self.stage_ms: Dict[str, float] = {}

    # ------------------------------------------------------------------ helpers
    def _peak(self, x: torch.Tensor, n: Sequence[int], eps: float) -> torch.Tensor:
        nd = torch.tensor(list(n), dtype=torch.int32, device=x.device)
        y = torch.empty_like(x)
        self.ctx.check(self.ctx.lib.ccx_peak_normalize(self.ctx.handle, x.data_ptr(), y.data_ptr(), x.shape[1], nd.data_ptr(), x.shape[0],
                                                       float(eps), _lib.current_stream_ptr()), "ccx_peak_normalize")
        return y

    def _cos(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """Row-wise cosine similarity of a [R, D] with b [R, D] (or a repeating [b_rows, D]): ccx_cosine_rows."""
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty(a.shape[0], device=a.device, dtype=torch.float32)
        self.ctx.check(self.ctx.lib.ccx_cosine_rows(self.ctx.handle, a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], b.shape[0],
                                                    out.data_ptr(), _lib.current_stream_ptr()), "ccx_cosine_rows")
        return out

    def _pad_batch(self, crops: List[torch.Tensor]):
        """Ragged 1-D device crops -> (padded [n, max_len] f32 buffer, lengths).  One gather launch (ccx_gather_rows)
        instead of one copy per crop; columns past a crop's length are uninitialised (the kernels read [:n] only)."""
        n = [int(c.numel()) for c in crops]
        dev = crops[0].device
        buf = torch.empty(len(crops), max(n), device=dev, dtype=torch.float32)
        for c in crops:
            assert c.is_cuda and c.dtype == torch.float32 and (c.dim() == 1 or c.is_contiguous()) and c.stride(-1) == 1
        ptrs = torch.tensor([c.data_ptr() for c in crops], dtype=torch.int64).to(dev, non_blocking=False)
        lens = torch.tensor(n, dtype=torch.int32).to(dev, non_blocking=False)
        self.ctx.check(self.ctx.lib.ccx_gather_rows(self.ctx.handle, ptrs.data_ptr(), lens.data_ptr(), len(crops), max(n), buf.data_ptr(),
                                                    buf.shape[1], _lib.current_stream_ptr()), "ccx_gather_rows")
        self._keep = (ptrs, lens)          # the tables must outlive the asynchronous launch
        return buf, n

    def _whisper(self, crops: List[torch.Tensor], prompts: List[str], prompt_ids: Optional[list] = None) -> List[dict]:
        """One 30 s window per crop (pinned schedule: every crop <= 30 s), batched in groups."""
        w = self.m["whisper_model"]
        tok = w.tokenizer
        out: List[dict] = []
        for i0 in range(0, len(crops), min(self.group, w.max_batch)):
            grp = crops[i0:i0 + min(self.group, w.max_batch)]
            buf, n = self._pad_batch(grp)                                               # kernels read only [:n_samples]
            w.log_mel(buf, n)
            w.encode(len(grp))
            pr = [w.initial_tokens(tok.encode(" " + p.strip()) if p else []) for p in prompts[i0:i0 + len(grp)]]
            if prompt_ids is not None:
                prompt_ids += pr
            out += w.decode_greedy(pr, sample_len=self.sample_len)
        return out

    def _mark(self, name: str, t0: float, timed: bool) -> float:
        if timed:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            self.stage_ms[name] = self.stage_ms.get(name, 0.0) + (t1 - t0) * 1e3
            return t1
        return t0

    # ------------------------------------------------------------------ the pinned pipeline
    def run_pinned(self, audio: torch.Tensor, timed: bool = False, debug: bool = False) -> Dict[str, object]:
        """audio: [B, 480000] f32 on the GPU (raw clips).  Returns token records + counters; with `debug` also every
        intermediate the parity test compares with the oracle-composed pipeline (tests/pinned_oracle.py): the gated clips,
        speaker profiles, segment / window similarities, separated sources, the picked source and the Whisper prompt ids.
        Sequential schedule: front end, then Whisper, on the caller's stream."""
        st = self._front(audio, timed, debug)
        prompt_ids: Optional[list] = [] if debug else None
        all_txt = self._whisper(st["whisper_crops"], st["whisper_prompts"], prompt_ids)
        return self._finish(st, all_txt, prompt_ids, timed)

    def _front(self, audio: torch.Tensor, timed: bool = False, debug: bool = False) -> dict:
        """Everything of the pinned pipeline before Whisper (gate, VAD, diarization, profiles, similarities, sliding windows,
        separation, source pick) on the current stream.  MFMA- / latency-bound; the Whisper decode that follows is HBM-bound."""
        m = self.m
        B, N = audio.shape
        assert N == 30 * SR and audio.is_cuda
        t = time.perf_counter()
        if timed:
            torch.cuda.synchronize(); t = time.perf_counter()
        # 1. load_audio: spectral gate on the whole clip + peak normalise (A3)
        den = self._peak(m["denoiser"].reduce_batch(audio, [N] * B, self.nra), [N] * B, 1e-8)
        t = self._mark("load_audio_gate", t, timed)
        # 2. VAD and diarization on the RAW clips (computed, not steering)
        items = [{"waveform": audio[b], "sample_rate": SR} for b in range(B)]
        vad_ann = m["vad_pipeline"].batch(items)
        n_vad = sum(len(a) for a in vad_ann)
        t = self._mark("vad", t, timed)
        diar_ann = m["diarization"].batch(items, min_speakers=1, max_speakers=2)
        n_diar = sum(len(a) for a in diar_ann)
        t = self._mark("diarization", t, timed)
        # 3. speaker profiles from the scheduled turns (all >= 0.75 s): gate each crop, normalise, embed (A8)
        sched = [(spk, int(s * SR), int(e * SR)) for spk, s, e in SCHEDULE_30S]
        crops = [den[b, s:e] for b in range(B) for _, s, e in sched]
        buf, n = self._pad_batch(crops)
        clean = self._peak(m["denoiser"].reduce_batch(buf, n, self.nra), n, 0.0)
        pe = m["embedding_model"].embed_batch([clean[i, :n[i]] for i in range(len(crops))])
        # embedding quality = unbiased variance of the raw crop (reference 939), then the variance-weighted sum of each speaker's turn
        # embeddings (not re-normalised, reference 946-953), all clips at once: ccx_row_variance + ccx_speaker_profiles
        nd = torch.tensor(n, device=buf.device, dtype=torch.int32)
        var = torch.empty(len(n), device=buf.device, dtype=torch.float32)
        self.ctx.check(self.ctx.lib.ccx_row_variance(self.ctx.handle, buf.data_ptr(), buf.shape[1], nd.data_ptr(), len(n), var.data_ptr(),
                                                     _lib.current_stream_ptr()), "ccx_row_variance")
        D = int(pe.shape[1])
        pe_c, var_c = pe.view(B, len(sched), D), var.view(B, len(sched))
        spk_ids = torch.tensor([0 if s_ == "A" else 1 for s_, _, _ in sched], device=buf.device, dtype=torch.int32)
        prof = torch.empty(B, 2, D, device=buf.device, dtype=torch.float32)
        self.ctx.check(self.ctx.lib.ccx_speaker_profiles(self.ctx.handle, pe.data_ptr(), var.data_ptr(), spk_ids.data_ptr(), B, len(sched), D, 2,
                                                         prof.data_ptr(), _lib.current_stream_ptr()), "ccx_speaker_profiles")
        prof_all = {"A": prof[:, 0], "B": prof[:, 1]}
        profiles = [{spk: prof_all[spk][b] for spk in ("A", "B")} for b in range(B)]
        t = self._mark("profiles", t, timed)
        # 4. regular segments (A 18-24, B 26-30): embed + similarity, then Whisper with the fixed prompt
        reg = [(b, spk, s, e) for b in range(B) for spk, s, e in sched[2:]]
        reg_crops = [den[b, s:e] for b, _, s, e in reg]
        re_ = m["embedding_model"].embed_batch(reg_crops)
        sims = self._cos(re_, torch.stack([profiles[b][spk] for b, spk, _, _ in reg]))
        t = self._mark("segment_embed", t, timed)
        # (their Whisper windows are decoded together with the overlap regions' at the end: the calls are independent,
        #  and one large decode batch amortises the latency-bound step chain)
        # 5. overlap-bearing segments (A 0-9, B 7-16): sliding-window attribution (0.8 s / 0.4 s)
        ov = [(b, spk, s, e) for b in range(B) for spk, s, e in sched[:2]]
        wins, owner = [], []
        for i, (b, _, s, e) in enumerate(ov):
            pos = s
            while pos + int(self.win * SR) <= e:
                wins.append(den[b, pos:pos + int(self.win * SR)])
                owner.append(i)
                pos += int(self.hop * SR)
        we = m["embedding_model"].embed_batch(wins)
        pa = torch.stack([profiles[ov[i][0]]["A"] for i in owner])
        pb = torch.stack([profiles[ov[i][0]]["B"] for i in owner])
        win_sims = torch.stack([self._cos(we, pa), self._cos(we, pb)], dim=1)   # [windows, 2]
        t = self._mark("sliding_windows", t, timed)
        # scripted window labels: each overlap segment splits at the scheduled overlap (7-9 s) into two regions
        regions = []
        for b, spk, s, e in ov:
            cut = int(7.0 * SR) if spk == "A" else int(9.0 * SR)
            regions += [(b, "A" if spk == "A" else "B", s, cut), (b, "B" if spk == "A" else "A", cut, e)]
        rcrops = [den[b, s:e] for b, _, s, e in regions]
        rn = [int(c.numel()) for c in rcrops]
        rbuf, _ = self._pad_batch(rcrops)
        sep = []
        i0 = 0
        while i0 < len(rcrops):        # greedy groups under the separator's token capacity (frames, padded to chunks)
            i1, tok = i0, 0
            while i1 < len(rcrops) and i1 - i0 < m["separator"].max_utts:
                need = m["separator"].tokens_for(rn[i1])
                if tok + need > m["separator"].max_tokens and i1 > i0:
                    break
                tok += need
                i1 += 1
            sep.append(m["separator"].separate_batch(rbuf[i0:i1], rn[i0:i1]))
            i0 = i1
        sep = torch.cat(sep, dim=0)                                       # [R, T, 2]
        t = self._mark("separate", t, timed)
        srcs = []
        for k in range(2):
            srcs.append(self._peak(sep[:, :, k].contiguous(), rn, 1e-8))
        se = [m["embedding_model"].embed_batch([srcs[k][i, :rn[i]] for i in range(len(rcrops))]) for k in range(2)]
        pr = torch.stack([profiles[b][spk] for b, spk, _, _ in regions])
        src_sims = torch.stack([self._cos(se[0], pr), self._cos(se[1], pr)], dim=1).cpu()     # [regions, 2]
        pick = (src_sims[:, 1] > src_sims[:, 0]).tolist()          # strict >: the first source wins a tie (reference 1094-1100)
        best = [srcs[1 if pick[i] else 0][i, :rn[i]] for i in range(len(regions))]
        t = self._mark("source_select", t, timed)
        st = dict(B=B, n_reg=len(reg_crops), whisper_crops=reg_crops + best,
                  whisper_prompts=[PROMPT_TWO_PEOPLE] * len(reg_crops) + [PROMPT_SINGLE] * len(best), n_vad=n_vad, n_diar=n_diar,
                  embeds=len(crops) + len(reg_crops) + len(wins) + 2 * len(rcrops), separator_calls=len(rcrops), sims=sims,
                  win_sims=win_sims, t=t)
        if debug:
            st["dbg"] = dict(den=den, pe_c=pe_c, var_c=var_c, prof_all=prof_all, owner=owner, sep=sep, rn=rn, se=se, pr=pr, pick=pick, src_sims=src_sims,
                             regions=regions, reg=reg, vad_ann=vad_ann, diar_ann=diar_ann)
        return st

    def _finish(self, st: dict, all_txt: List[dict], prompt_ids: Optional[list], timed: bool) -> Dict[str, object]:
        """Assemble run_pinned's result from the front-end state and the decoded records."""
        B, n_reg = st["B"], st["n_reg"]
        reg_crops, best = st["whisper_crops"][:n_reg], st["whisper_crops"][n_reg:]
        sims, win_sims = st["sims"], st["win_sims"]
        reg_txt, ov_txt = all_txt[:n_reg], all_txt[n_reg:]
        self._mark("whisper", st["t"], timed)
        dbg = {}
        debug = "dbg" in st
        if debug:
            d = st["dbg"]
            den, pe_c, var_c, prof_all, owner, sep, rn, se, pr, pick, regions, reg = (d[k] for k in (
                "den", "pe_c", "var_c", "prof_all", "owner", "sep", "rn", "se", "pr", "pick", "regions", "reg"))
        if debug:
            dbg = dict(den=den.cpu(), profile_embeds=pe_c.cpu(), profile_var=var_c.cpu(), profiles={k: v.cpu() for k, v in prof_all.items()},
                       window_sims_full=win_sims.cpu(), window_owner=list(owner), separated=sep.cpu(), region_len=list(rn),
                       source_sims=d["src_sims"],
                       pick=[int(x) for x in pick], prompt_ids=prompt_ids, whisper_inputs=[c.cpu() for c in reg_crops + best],
                       regions=[(b, spk, s, e) for b, spk, s, e in regions], regular=[(b, spk, s, e) for b, spk, s, e in reg],
                       vad=[[(sg.start, sg.end) for sg, _ in a.itertracks()] for a in d["vad_ann"]],
                       diarization=[[(sg.start, sg.end, l) for sg, _, l in a.itertracks(yield_label=True)] for a in d["diar_ann"]])
        return dict(**dbg, n_clips=B, audio_seconds=30.0 * B, whisper_calls=len(reg_crops) + len(best), tokens=sum(len(r["tokens"]) for r in reg_txt + ov_txt),
                    embeds=st["embeds"], separator_calls=st["separator_calls"],
                    vad_regions=st["n_vad"], diar_turns=st["n_diar"], records=reg_txt + ov_txt, sims=sims.cpu().tolist(), window_sims=int(win_sims.shape[0]))


    # ------------------------------------------------------------------ software-pipelined schedule
    def run_pinned_pipelined(self, audios: Sequence[torch.Tensor], debug: bool = False, span: int = 1) -> List[Dict[str, object]]:
        """The same work as [run_pinned(a) for a in audios], software-pipelined across batches: while the Whisper windows of
        batch i decode (HBM-bound cross attention + latency-bound chain, on the decode lanes' own high-priority streams, driven
        by a worker thread -- the C call releases the GIL), the front end and the encoder of batch i + 1 (MFMA-bound) run on a
        second stream.  Two Whisper instances (models["whisper_models"]) alternate, so an encode never overwrites cross-KV that
        is still being decoded.  Results are identical to the sequential schedule: same kernels, same order per batch.

        `span` > 1: the Whisper windows of `span` consecutive batches are encoded and decoded TOGETHER (one decode group of
        span x 6 x B sequences).  The decode step is a latency-bound chain of ~150 small kernels per lane plus the HBM-bound
        cross attention; a lane's chain hides under the other lanes' cross attention only when that cross attention lasts as
        long as the chain, i.e. with enough sequences per lane ("Decode lanes")."""
        from concurrent.futures import ThreadPoolExecutor
        whs = self.m.get("whisper_models") or [self.m["whisper_model"]]
        if len(whs) < 2:
            raise _lib.CcxError("run_pinned_pipelined needs two Whisper instances: load_models(..., whisper_instances=2)")
        if self.ctx.prof_on:
            raise _lib.CcxError("run_pinned_pipelined: switch the per-launch profile off (ctx.prof_enable(False)) -- it is recorded per "
                                "context, and this schedule drives the context from two host threads")
        dev = audios[0].device
        if not hasattr(self, "_front_stream"):
            self._front_stream = torch.cuda.Stream(device=dev)
            self._dec_streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in whs]     # lane 0 of each instance
            torch.cuda.synchronize(dev)
            for w, s_ in zip(whs, self._dec_streams):       # the lane probe needs an idle device: do it before anything overlaps
                w.prepare_lanes(s_)
        torch.cuda.current_stream(dev).synchronize()      # the inputs were produced on the caller's stream
        grp = min(self.group, whs[0].max_batch)

        # optional list: (what, batch unit, t_start, t_end) -- decodes in host time of the worker thread, front ends and encodes as
        # stream events (resolved to the same clock after the run: the front thread is never synchronised for the trace)
        trace = getattr(self, "trace", None)
        ev_trace = []
        if trace is not None:
            self._front_stream.synchronize()
            ev_base = torch.cuda.Event(enable_timing=True); ev_base.record(self._front_stream); ev_base.synchronize()
            t_base = time.perf_counter()

        def stamp():
            e = torch.cuda.Event(enable_timing=True); e.record(self._front_stream)
            return e

        def decode_task(k: int, prompts, ready: torch.cuda.Event, u: int):
            with torch.cuda.stream(self._dec_streams[k]):
                self._dec_streams[k].wait_event(ready)
                if trace is not None:
                    ready.synchronize(); t0 = time.perf_counter()
                r = whs[k].decode_greedy(prompts, sample_len=self.sample_len)
                if trace is not None:
                    trace.append(("decode", u, t0, time.perf_counter()))
                return r

        out: List[Dict[str, object]] = []
        pending = []                       # (front state, [futures], prompt ids)
        slot_busy = [None] * len(whs)
        unit = 0
        with ThreadPoolExecutor(max_workers=1) as ex, torch.cuda.stream(self._front_stream):
            held = []                           # front states waiting for the rest of their span
            for ai, audio in enumerate(audios):
                e0 = stamp() if trace is not None else None
                th0 = time.perf_counter()
                held.append(self._front(audio, False, debug))
                if trace is not None:
                    ev_trace.append(("front", unit, e0, stamp()))
                    trace.append(("front-host", unit, th0, time.perf_counter()))
                if len(held) < max(1, span) and ai + 1 < len(audios):
                    continue
                # one Whisper unit for the windows of every held batch
                crops = [c for h_ in held for c in h_["whisper_crops"]]
                prompts = [q for h_ in held for q in h_["whisper_prompts"]]
                bounds = np.cumsum([0] + [len(h_["whisper_crops"]) for h_ in held]).tolist()
                futs, pids = [], []
                for i0 in range(0, len(crops), grp):
                    k = unit % len(whs)
                    if slot_busy[k] is not None:
                        slot_busy[k].result()                      # the instance's previous windows are decoded
                    w = whs[k]
                    e0 = stamp() if trace is not None else None
                    part = crops[i0:i0 + grp]
                    buf, n = self._pad_batch(part)
                    w.log_mel(buf, n)
                    w.encode(len(part))
                    ready = torch.cuda.Event()
                    ready.record(self._front_stream)
                    if trace is not None:
                        ev_trace.append(("encode", unit, e0, stamp()))
                    tok = w.tokenizer
                    pr = [w.initial_tokens(tok.encode(" " + p.strip()) if p else []) for p in prompts[i0:i0 + len(part)]]
                    pids += pr
                    fut = ex.submit(decode_task, k, pr, ready, unit)
                    slot_busy[k] = fut
                    futs.append((fut, buf))                        # buf stays referenced until its kernels have run
                    unit += 1
                pending.append((held, bounds, futs, pids))
                held = []
                # hand finished units over as soon as their decodes are done (keeps at most two units of state alive)
                while len(pending) > 2:
                    out += self._finish_unit(*pending.pop(0), debug)
            for u in pending:
                out += self._finish_unit(*u, debug)
            self._front_stream.synchronize()
        for what, u, e0, e1 in ev_trace:
            trace.append((what, u, t_base + ev_base.elapsed_time(e0) * 1e-3, t_base + ev_base.elapsed_time(e1) * 1e-3))
        return out

    def _finish_unit(self, held, bounds, futs, pids, debug) -> List[Dict[str, object]]:
        recs = [r for f, _ in futs for r in f.result()]
        return [self._finish(st, recs[bounds[k]:bounds[k + 1]], pids[bounds[k]:bounds[k + 1]] if debug else None, False)
                for k, st in enumerate(held)]


def shard_clip_indices(n_clips: int, rank: int, world: int) -> List[int]:
    """clip i -> rank i mod world: independent units, no data-path collective."""
    return [i for i in range(n_clips) if i % world == rank]


def broadcast_weights(state_dicts: Optional[dict], src: int = 0, device="cpu", force_collective: bool = False) -> dict:
    """The start-of-job collective (C1): rank `src` holds the weights of every model
    (models.build_state_dicts); all tensors are packed into ONE byte blob and sent with ONE broadcast (RCCL: a direct
    1 -> N-1 send over the xGMI links), the small manifest (names / shapes / dtypes / non-tensor entries) as an object
    broadcast before it.  Returns the same nested dict on every rank (host tensors: views of one host copy of the received blob).
    `force_collective`: pack, broadcast and unpack even in a one-rank group (lets RCCL run this code on a single GPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        if state_dicts is None:
            raise ValueError("broadcast_weights: no process group and no local weights")
        return state_dicts
    rank = dist.get_rank()
    manifest, parts, off = None, [], 0
    if rank == src:
        if state_dicts is None:
            raise ValueError("broadcast_weights: the source rank needs the weights")
        manifest = {"plain": {}, "tensors": []}
        for model, sd in state_dicts.items():
            if not isinstance(sd, dict) or not any(torch.is_tensor(v) for v in sd.values()):
                manifest["plain"][model] = sd
                continue
            for key, t in sd.items():
                if not torch.is_tensor(t):
                    manifest["tensors"].append((model, key, None, None, None, 0, t))
                    continue
                t = t.detach().contiguous().cpu()
                nbytes = t.numel() * t.element_size()
                manifest["tensors"].append((model, key, tuple(t.shape), str(t.dtype).replace("torch.", ""), off, nbytes, None))
                parts.append(t.reshape(-1).view(torch.uint8))
                pad = -nbytes % 16                     # keep every tensor 16-byte aligned inside the blob
                if pad:
                    parts.append(torch.zeros(pad, dtype=torch.uint8))
                off += nbytes + pad
        manifest["total"] = off
    box = [manifest]
    dist.broadcast_object_list(box, src=src)
    manifest = box[0]
    if rank == src:
        blob = (torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8)).to(device)
    else:
        blob = torch.empty(manifest["total"], dtype=torch.uint8, device=device)
    if manifest["total"]:
        dist.broadcast(blob, src=src)
    host = blob.cpu()        # state dicts are host objects everywhere else (loaders repack them before the upload)
    out: dict = dict(manifest["plain"])
    for model, key, shape, dtype, o, nbytes, plain in manifest["tensors"]:
        d = out.setdefault(model, {})
        d[key] = plain if shape is None else host[o:o + nbytes].view(getattr(torch, dtype)).reshape(shape)
    return out


def gather_transcripts(records: List[dict], sample_len: int, eot: int, device) -> Optional[torch.Tensor]:
    """All-gather of fixed-size token records over RCCL (xGMI): the one data-path collective (C2)."""
    import torch.distributed as dist
    rec = torch.full((len(records), sample_len + 1), eot, dtype=torch.int32, device=device)
    for i, r in enumerate(records):
        rec[i, 0] = len(r["tokens"])
        if r["tokens"]:
            rec[i, 1:1 + len(r["tokens"])] = torch.tensor(r["tokens"], dtype=torch.int32, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return rec
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.cat(out, dim=0)
