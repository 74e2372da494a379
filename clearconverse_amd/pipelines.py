"""VAD and speaker-diarization pipelines on top of the libccx networks.

Drop-ins for `self.vad_pipeline(path)` and `self.diarization(path, min_speakers=, max_speakers=)`
of the reference (Pipeline.from_pretrained("pyannote/voice-activity-detection") and
("pyannote/speaker-diarization-3.1"), /root/reference/back/api.py:782-792; called at back/api.py:1311,
1056-1060, 1124-1128).  Both return an `Annotation` whose `itertracks(yield_label=True)` yields
(segment, track, label) with float `segment.start/.end` -- the only API the reference touches.

The networks (SincNet/PyanNet segmentation, WeSpeaker ResNet-34 embedder of speaker-diarization-3.1) run in libccx;
this module is the small host-side post-net (K22 in SURVEY.md): sliding windows, powerset decoding, overlap-add
aggregation, hysteresis binarisation, agglomerative clustering, timeline reconstruction, restated from
recollection of pyannote.audio 3.x [UPSTREAM-RECALL].  Hyper-parameters come from the pipelines' own config.yaml when it is
on disk (weights.find_pipeline_config, wired in models.load_models); the constructor defaults are the recalled values of the
published configs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .audio import SincResampler, read_wav

SR = 16000
FRAME_STEP = 270 / SR          # SincNet: stride 10 x three maxpool3
FRAME_DUR = 991 / SR           # receptive field of one output frame


class Segment:
    __slots__ = ("start", "end")

    def __init__(self, start: float, end: float):
        self.start, self.end = float(start), float(end)

    @property
    def duration(self):
        return self.end - self.start

    def __repr__(self):
        return f"[{self.start:.3f} --> {self.end:.3f}]"


class Annotation:
    """Minimal stand-in for pyannote.core.Annotation (tracks sorted by start time)."""

    def __init__(self, tracks: Sequence[Tuple[float, float, str]] = ()):
        self._tracks = sorted(((float(s), float(e), str(l)) for s, e, l in tracks), key=lambda t: (t[0], t[1]))

    def itertracks(self, yield_label: bool = False):
        for i, (s, e, l) in enumerate(self._tracks):
            yield (Segment(s, e), f"T{i}", l) if yield_label else (Segment(s, e), f"T{i}")

    def labels(self):
        return sorted({l for _, _, l in self._tracks})

    def __len__(self):
        return len(self._tracks)


def load_mono_16k(path_or_wave):
    """The pipelines read the file themselves (raw, NOT denoised -- SURVEY.md section 3b)."""
    if isinstance(path_or_wave, dict):
        w = path_or_wave["waveform"]
        sr = int(path_or_wave.get("sample_rate", SR))
        if torch.is_tensor(w) and w.is_cuda and sr == SR and w.dtype == torch.float32 and (w.dim() == 1 or w.shape[0] == 1):
            return w.reshape(-1)            # already resident: the windows are cut on the device, no host round trip
        x = w.detach().cpu().numpy() if torch.is_tensor(w) else np.asarray(w)
    else:
        x, sr = read_wav(str(path_or_wave))
    x = np.asarray(x, dtype=np.float32)
    if x.ndim > 1:
        x = x.mean(axis=0)
    if sr != SR:
        return SincResampler(sr, SR)(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)))     # K1 on the device, stays resident
    return np.ascontiguousarray(x, dtype=np.float32)


def sliding_chunks(n: int, win: int, step: int) -> List[int]:
    """Chunk start samples: every `step`, plus one final chunk (zero padded) covering the tail."""
    if n <= win:
        return [0]
    starts = list(range(0, n - win + 1, step))
    if starts[-1] + win < n:
        starts.append(starts[-1] + step)
    return starts


def powerset_to_multilabel(logp: np.ndarray, n_spk: int = 3, max_set: int = 2) -> np.ndarray:
    """[frames, 7] powerset log-probs -> hard [frames, 3] multi-label (argmax class -> its speaker set).
    Class order: empty, {0}, {1}, {2}, {0,1}, {0,2}, {1,2}."""
    sets: List[Tuple[int, ...]] = [()]
    import itertools
    for k in range(1, max_set + 1):
        sets += list(itertools.combinations(range(n_spk), k))
    table = np.zeros((len(sets), n_spk), dtype=np.float32)
    for i, s in enumerate(sets):
        table[i, list(s)] = 1.0
    return table[np.argmax(logp, axis=-1)]


def aggregate(chunks: Sequence[np.ndarray], starts: Sequence[int], n_samples: int, win: int) -> np.ndarray:
    """Overlap-add average of per-chunk frame scores [frames, C] onto one global frame grid."""
    C = chunks[0].shape[1]
    n_out = int(np.ceil(n_samples / 270)) + 1
    acc = np.zeros((n_out, C), dtype=np.float64)
    cnt = np.zeros((n_out, 1), dtype=np.float64)
    for sc, s0 in zip(chunks, starts):
        f0 = int(round(s0 / 270))
        f1 = min(n_out, f0 + sc.shape[0])
        acc[f0:f1] += sc[: f1 - f0]
        cnt[f0:f1] += 1
    return (acc / np.maximum(cnt, 1)).astype(np.float32)


def aggregate_cm(cm: np.ndarray, starts: Sequence[int], n_samples: int) -> np.ndarray:
    """`aggregate` for equal-sized windows held class-major: cm [C, windows, frames] -> [n_out, C].  Same accumulation
    order and precision as `aggregate`; the class axis is outermost so every slice added is contiguous (reductions
    and slices along a trailing axis of 3 or 7 entries are what numpy is slowest at)."""
    C, _, F = cm.shape
    n_out = int(np.ceil(n_samples / 270)) + 1
    acc = np.zeros((C, n_out), dtype=np.float64)
    cnt = np.zeros(n_out, dtype=np.float64)
    for w, s0 in enumerate(starts):
        f0 = int(round(s0 / 270))
        f1 = min(n_out, f0 + F)
        acc[:, f0:f1] += cm[:, w, : f1 - f0]
        cnt[f0:f1] += 1
    return np.ascontiguousarray((acc / np.maximum(cnt, 1)).T.astype(np.float32))


def _powerset_table(n_spk: int = 3, max_set: int = 2) -> np.ndarray:
    import itertools
    sets: List[Tuple[int, ...]] = [()]
    for k in range(1, max_set + 1):
        sets += list(itertools.combinations(range(n_spk), k))
    table = np.zeros((len(sets), n_spk), dtype=np.float32)
    for i, st in enumerate(sets):
        table[i, list(st)] = 1.0
    return table


_POWERSET_T = np.ascontiguousarray(_powerset_table().T)      # [speaker, class]


def multilabel_cm(arr: np.ndarray, powerset: bool) -> np.ndarray:
    """[windows, frames, classes] scores -> hard multi-label activity, class-major [speakers, windows, frames] float32
    (powerset: argmax class -> its speaker set, as `powerset_to_multilabel`; multi-label: score > 0.5)."""
    cm = np.ascontiguousarray(np.moveaxis(arr, -1, 0))
    if not powerset:
        return (cm > 0.5).astype(np.float32)
    idx = np.argmax(cm, axis=0)                                  # reduction over the OUTER axis: vectorised over frames
    return np.stack([np.take(_POWERSET_T[sp], idx) for sp in range(_POWERSET_T.shape[0])])


def binarize(score: np.ndarray, onset: float, offset: float, min_on: float = 0.0, min_off: float = 0.0,
             frame_step: float = FRAME_STEP, t0: float = 0.5 * FRAME_DUR) -> List[Tuple[float, float]]:
    """Hysteresis thresholding of a 1-D frame score (pyannote Binarize): on above `onset`, off below `offset`;
    then fill gaps < min_off and drop regions < min_on.  Frame i is centred at t0 + i*frame_step."""
    score = np.asarray(score)
    times = t0 + frame_step * np.arange(score.shape[0])
    regions: List[Tuple[float, float]] = []
    if score.shape[0] and onset >= offset:
        # vectorised state machine: +1 events switch on, -1 events switch off, every other frame keeps the state of the
        # last event before it (forward fill); a region starts at the frame that switches on and ends at the frame that
        # switches off (or at the last frame)
        ev = np.where(score > onset, 1, np.where(score < offset, -1, 0)).astype(np.int8)
        last = np.maximum.accumulate(np.where(ev != 0, np.arange(ev.shape[0]), -1))
        active = np.where(last >= 0, ev[np.maximum(last, 0)], -1) == 1
        d = np.diff(np.concatenate([[False], active]).astype(np.int8))
        starts, ends = np.flatnonzero(d == 1), np.flatnonzero(d == -1)
        for k, s_i in enumerate(starts):
            regions.append((float(times[s_i]), float(times[ends[k]] if k < len(ends) else times[-1])))
    else:
        active, start = False, 0.0
        for t, y in zip(times, score):
            if active:
                if y < offset:
                    regions.append((start, float(t)))
                    active = False
            elif y > onset:
                start, active = float(t), True
        if active:
            regions.append((start, float(times[-1])))
    merged: List[Tuple[float, float]] = []
    for s, e in regions:
        if merged and s - merged[-1][1] < min_off:
            merged[-1] = (merged[-1][0], e)
        else:
            merged.append((s, e))
    return [(s, e) for s, e in merged if e - s >= min_on and e > s]


class VoiceActivityDetection:
    """pyannote/voice-activity-detection: multi-label segmentation (sigmoid, 3 speakers), 5 s windows with
    10 % step, max over speakers, overlap-add average, hysteresis binarisation."""

    def __init__(self, seg_net, duration: float = 5.0, step_ratio: float = 0.1, onset: float = 0.767, offset: float = 0.377,
                 min_duration_on: float = 0.136, min_duration_off: float = 0.067, batch: int = 64):
        self.net = seg_net
        self.win, self.step = int(duration * SR), int(duration * step_ratio * SR)
        self.onset, self.offset, self.min_on, self.min_off, self.batch_size = onset, offset, min_duration_on, min_duration_off, batch

    def _chunks(self, x: np.ndarray):
        starts = sliding_chunks(len(x), self.win, self.step)
        dev = torch.from_numpy(x).to(self.net.device) if not torch.is_tensor(x) else x.to(self.net.device)
        crops = []
        for s in starts:
            c = dev[s:s + self.win]
            if c.numel() < self.win:
                c = torch.nn.functional.pad(c, (0, self.win - c.numel()))
            crops.append(c)
        return starts, crops

    def _score(self, outs, starts, n):
        if len({o.shape for o in outs}) == 1:
            outs = np.stack(outs)               # equal windows: class-major, one contiguous add per window
            if self.net.powerset:
                sc = 1.0 - np.exp(outs[..., 0])
            else:
                sc = outs[..., 0]
                for c in range(1, outs.shape[-1]):
                    sc = np.maximum(sc, outs[..., c])
            return aggregate_cm(sc[None], starts, n)[:, 0]
        if self.net.powerset:   # a powerset model used as VAD: speech = 1 - P(empty set)
            sc = [1.0 - np.exp(o[:, :1]) for o in outs]
        else:
            sc = [o.max(axis=-1, keepdims=True) for o in outs]
        return aggregate(sc, starts, n, self.win)[:, 0]

    def begin(self, items: Sequence):
        """Cut the windows of all items and QUEUE the network; returns a handle for `finish`.  Nothing waits for the GPU
        here, so a caller can queue more work (e.g. the diarization network) before collecting."""
        xs = [load_mono_16k(it) for it in items]
        plans, crops = [], []
        for x in xs:
            if len(x) < 991 * 4:
                plans.append(None)
                continue
            starts, cr = self._chunks(x)
            plans.append((starts, len(crops), len(cr)))
            crops += cr
        return xs, plans, (self.net.segment_launch(crops) if crops else [])

    def finish(self, handle) -> List[Annotation]:
        xs, plans, pending = handle
        outs = self.net.segment_fetch(pending)
        anns = []
        for x, pl in zip(xs, plans):
            if pl is None:
                anns.append(Annotation([]))
                continue
            starts, i0, n = pl
            score = self._score(outs[i0:i0 + n], starts, len(x))
            n_valid = min(len(score), int(len(x) / 270))
            regions = binarize(score[:n_valid], self.onset, self.offset, self.min_on, self.min_off)
            dur = len(x) / SR
            anns.append(Annotation([(max(0.0, s), min(dur, e), "SPEECH") for s, e in regions if min(dur, e) > max(0.0, s)]))
        return anns

    def batch(self, items: Sequence) -> List[Annotation]:
        """Several files / waveforms at once: all windows of all items go through the network together."""
        return self.finish(self.begin(items))

    def __call__(self, path_or_wave) -> Annotation:
        return self.batch([path_or_wave])[0]


def agglomerative_centroid(emb: np.ndarray, threshold: float, min_cluster_size: int, min_clusters: int, max_clusters: int) -> np.ndarray:
    """pyannote AgglomerativeClustering (centroid linkage on unit-normalised embeddings, euclidean): cut at
    `threshold`, keep clusters with >= min_cluster_size members as "large", re-assign the rest to the nearest
    large centroid, then force the number of clusters into [min_clusters, max_clusters]."""
    from scipy.cluster.hierarchy import fcluster, linkage
    n = emb.shape[0]
    if n == 1:
        return np.zeros(1, dtype=np.int64)
    e = emb / np.maximum(np.linalg.norm(emb, axis=1, keepdims=True), 1e-12)
    Z = linkage(e, method="centroid", metric="euclidean")
    lab = fcluster(Z, threshold, criterion="distance") - 1
    sizes = np.bincount(lab)
    big = min(min_cluster_size, max(1, n // 4))
    large = [c for c in range(len(sizes)) if sizes[c] >= big]
    if not large:
        large = [int(np.argmax(sizes))]
    if len(large) > max_clusters:
        large = sorted(large, key=lambda c: -sizes[c])[:max_clusters]
    elif len(large) < min_clusters:
        # centroid linkage is not monotonic, so "maxclust = k" may return fewer than k clusters: grow the
        # request until at least min_clusters come out, then keep the min_clusters largest
        for m in range(min_clusters, n + 1):
            lab = fcluster(Z, m, criterion="maxclust") - 1
            sizes = np.bincount(lab)
            if np.count_nonzero(sizes) >= min(min_clusters, n):
                break
        large = sorted([c for c in range(len(sizes)) if sizes[c] > 0], key=lambda c: -sizes[c])[:min_clusters]
    cents = np.stack([e[lab == c].mean(axis=0) for c in large])
    d = ((e[:, None, :] - cents[None, :, :]) ** 2).sum(-1)
    return np.argmin(d, axis=1).astype(np.int64)


class SpeakerDiarization:
    """pyannote/speaker-diarization-3.1 shape: powerset segmentation over 10 s windows (10 % step), one
    embedding per (window, local speaker) from overlap-free frames, agglomerative clustering, per-frame
    speaker count x clustered activations -> timeline."""

    def __init__(self, seg_net, embedder, duration: float = 10.0, step_ratio: float = 0.1, threshold: float = 0.7045654963945799,
                 min_cluster_size: int = 12, min_duration_off: float = 0.0, min_active_ratio: float = 0.2, batch: int = 32):
        self.net, self.emb = seg_net, embedder
        self.win, self.step = int(duration * SR), int(duration * step_ratio * SR)
        self.threshold, self.min_cluster_size, self.min_off, self.min_active, self.batch_size = threshold, min_cluster_size, min_duration_off, min_active_ratio, batch

    def __call__(self, path_or_wave, min_speakers: Optional[int] = None, max_speakers: Optional[int] = None,
                 num_speakers: Optional[int] = None) -> Annotation:
        return self.batch([path_or_wave], min_speakers, max_speakers, num_speakers)[0]

    def batch(self, items: Sequence, min_speakers: Optional[int] = None, max_speakers: Optional[int] = None,
              num_speakers: Optional[int] = None) -> List[Annotation]:
        """Several files / waveforms at once: segmentation windows and speaker-embedding crops of all items
        are batched through the networks; clustering / reconstruction stay per item."""
        return self.finish(self.embed(self.begin(items)), min_speakers, max_speakers, num_speakers)

    # The three stages of `batch`, separately callable so that a driver can keep the GPU busy while the host works:
    # begin  -- cut windows, QUEUE the segmentation network (no wait);
    # embed  -- fetch the frame scores, pick the local speakers' pooling masks on the host, QUEUE the embedding network;
    # finish -- fetch the embeddings, cluster and rebuild the timelines on the host.
    def begin(self, items: Sequence):
        xs = [load_mono_16k(it) for it in items]
        plans, crops = [], []
        for x in xs:
            if len(x) < 991 * 4:
                plans.append(None)
                continue
            starts = sliding_chunks(len(x), self.win, self.step)
            dev = x.to(self.net.device) if torch.is_tensor(x) else torch.from_numpy(x).to(self.net.device)
            cr = []
            for s in starts:
                c = dev[s:s + self.win]
                if c.numel() < self.win:
                    c = torch.nn.functional.pad(c, (0, self.win - c.numel()))
                cr.append(c)
            plans.append((starts, len(crops), len(cr)))
            crops += cr
        return xs, plans, crops, (self.net.segment_launch(crops) if crops else [])

    def embed(self, handle):
        xs, plans, crops, pending = handle
        seg = self.net.segment_fetch(pending)
        # local speakers of every window of every item
        per_item = []
        e_crops, e_weights = [], []
        for x, pl in zip(xs, plans):
            if pl is None:
                per_item.append(None)
                continue
            starts, i0, n = pl
            arr = np.stack(seg[i0:i0 + n])                                   # [windows, frames, classes]: windows are equal-sized
            mc = multilabel_cm(arr, self.net.powerset)                       # [speakers, windows, frames] float32 0/1
            n_spk = mc.sum(axis=0)                                           # active speakers per frame
            alone = n_spk == 1
            cleanf = mc * alone                                              # overlap-free activity
            n_act, n_clean = mc.sum(axis=-1), cleanf.sum(axis=-1)            # [speakers, windows] frame counts (exact in fp32)
            keep = n_act.astype(np.float64) / mc.shape[-1] >= self.min_active  # local speakers active for >= min_active_ratio
            use_clean = n_clean >= 0.5 * n_act                               # prefer overlap-free frames when enough remain
            kw, ks = np.nonzero(keep.T)                                      # row-major: window, then speaker
            keys = list(zip(kw.tolist(), ks.tolist()))
            e_crops += (i0 + kw).tolist()
            if len(keys):
                e_weights.append(np.where(use_clean[ks, kw][:, None], cleanf[ks, kw], mc[ks, kw]))   # [keys, frames] pooling masks
            per_item.append((starts, mc, keys, n_spk))
        if not e_crops:
            embs = None
        elif hasattr(self.emb, "embed_chunks"):
            # chunk-level embedder (WeSpeaker ResNet-34): the trunk runs once per window, pooling once per local speaker
            used = sorted(set(e_crops))
            where = {g: k for k, g in enumerate(used)}
            embs = self.emb.embed_chunks(torch.stack([crops[g] for g in used]), torch.from_numpy(np.concatenate(e_weights)), [where[g] for g in e_crops])
        else:
            embs = self.emb.embed_batch([crops[g] for g in e_crops], weights=list(torch.from_numpy(np.concatenate(e_weights))))
        return xs, per_item, embs                                            # embs: device tensor, still being computed

    def finish(self, handle, min_speakers: Optional[int] = None, max_speakers: Optional[int] = None,
               num_speakers: Optional[int] = None) -> List[Annotation]:
        lo = num_speakers or min_speakers or 1
        hi = num_speakers or max_speakers or 20
        xs, per_item, embs = handle
        embs = np.zeros((0, getattr(self.emb, "DIM", 512)), dtype=np.float32) if embs is None else embs.cpu().numpy()
        anns, e0 = [], 0
        for x, it in zip(xs, per_item):
            if it is None or not it[2]:
                anns.append(Annotation([]))
                continue
            starts, mc, keys, n_spk = it
            anns.append(self._reconstruct(len(x), starts, mc, keys, embs[e0:e0 + len(keys)], lo, hi, n_spk))
            e0 += len(keys)
        return anns

    def _reconstruct(self, n_samples: int, starts, mc, keys, embs, lo: int, hi: int, n_spk=None) -> Annotation:
        """mc: hard local activity [speakers, windows, frames]; keys: the (window, local speaker) pairs that were embedded."""
        dur = n_samples / SR
        if n_spk is None:
            n_spk = mc.sum(axis=0)
        count = np.rint(aggregate_cm(n_spk[None], starts, n_samples)[:, 0]).astype(np.int64)
        count = np.minimum(count, hi)
        ok = np.isfinite(embs).all(axis=1)
        labels = np.full(len(keys), -1, dtype=np.int64)
        if ok.any():
            labels[ok] = agglomerative_centroid(embs[ok], self.threshold, self.min_cluster_size, lo, hi)
        n_clusters = int(labels.max()) + 1
        if n_clusters <= 0:
            return Annotation([])
        clustered = np.zeros((n_clusters,) + mc.shape[1:], dtype=np.float32)
        for (kc, sp), lab in zip(keys, labels):                            # one pass over the (window, local speaker) keys
            if lab >= 0:
                np.maximum(clustered[lab, kc], mc[sp, kc], out=clustered[lab, kc])
        agg = aggregate_cm(clustered, starts, n_samples)                   # [frames, clusters]
        n_valid = min(agg.shape[0], int(n_samples / 270))
        agg, count = agg[:n_valid], count[:n_valid]
        # to_diarization: at each frame the `count` most active clusters speak
        order = np.argsort(-agg, axis=1)
        ranks = np.argsort(order, axis=1)                                  # rank of every cluster at every frame
        binary = ((ranks < count[:, None]) & (agg > 0)).astype(np.float32)
        tracks = []
        first_seen = {}
        for c in range(n_clusters):
            regs = binarize(binary[:, c], 0.5, 0.5, 0.0, self.min_off)
            for s, e in regs:
                s, e = max(0.0, s - 0.5 * FRAME_STEP), min(dur, e + 0.5 * FRAME_STEP)
                if e > s:
                    tracks.append((s, e, c))
                    first_seen[c] = min(first_seen.get(c, 1e9), s)
        rename = {c: f"SPEAKER_{i:02d}" for i, c in enumerate(sorted(first_seen, key=lambda c: first_seen[c]))}
        return Annotation([(s, e, rename[c]) for s, e, c in tracks])
