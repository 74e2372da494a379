"""VAD and speaker-diarization pipelines on top of the libccx networks.

Drop-ins for `self.vad_pipeline(path)` and `self.diarization(path, min_speakers=, max_speakers=)`
of the reference (Pipeline.from_pretrained("pyannote/voice-activity-detection") and
("pyannote/speaker-diarization-3.1"), /root/reference/back/api.py:782-792; called at back/api.py:1311,
1056-1060, 1124-1128).  Both return an `Annotation` whose `itertracks(yield_label=True)` yields
(segment, track, label) with float `segment.start/.end` -- the only API the reference touches.

The networks (SincNet/PyanNet segmentation, WeSpeaker ResNet-34 embedder of speaker-diarization-3.1) run in libccx;
this module is the small host-side post-net (K22 in SURVEY.md): sliding windows, powerset decoding, overlap-add
aggregation, hysteresis binarisation, agglomerative clustering, timeline reconstruction, restated from
recollection of pyannote.audio 3.1 [UPSTREAM-RECALL]; the frame-by-frame restatement it is tested against is
oracle/pyannote_pipeline_ref.py (parity unpinned: no reference fixture exists for these pipelines).  Hyper-parameters come from the pipelines' own config.yaml when it is
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


_RESAMPLERS: Dict[Tuple[int, int], SincResampler] = {}      # (source rate, device index) -> resampler (filter table uploaded once)


def load_mono_16k(path_or_wave, device: Optional[torch.device] = None, ctx=None):
    """The pipelines read the file themselves (raw, NOT denoised -- SURVEY.md section 3b).  Returns a 1-D float32 signal at
    16 kHz: a tensor on `device` when it is already resident there or had to be resampled (K1 runs on the device), a host numpy
    array otherwise (the callers move their windows to the device themselves)."""
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
        idx = device.index if isinstance(device, torch.device) and device.index is not None else 0
        # keyed by the context too, and an entry whose context was closed is rebuilt (Context.close() sets handle = None)
        key = (sr, idx, id(ctx))
        rs = _RESAMPLERS.get(key)
        if rs is None or getattr(getattr(rs, "ctx", None), "handle", True) is None:
            rs = _RESAMPLERS[key] = SincResampler(sr, SR, device=idx, ctx=ctx)
        return rs(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)))     # K1 on the pipeline's device, stays resident
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


def n_frames_out(n_chunks: int, win: int, step: int) -> int:
    """Length of the global frame grid of `n_chunks` windows (Inference.aggregate: closest frame of the last window's end, + 1)."""
    return int(np.rint((win + (n_chunks - 1) * step) / 270.0)) + 1


def aggregate(chunks: Sequence[np.ndarray], starts: Sequence[int], win: int, step: int, hamming: bool = False,
              average: bool = True) -> np.ndarray:
    """Overlap-add of per-chunk frame scores [frames, C] onto one global frame grid (Inference.aggregate, warm-up 0): flat or
    Hamming-weighted, averaged or summed; float32 accumulators in window order, frames no window covers are 0."""
    return aggregate_cm(np.ascontiguousarray(np.moveaxis(np.stack(chunks), -1, 0)), starts, win, step, hamming, average)


def aggregate_cm(cm: np.ndarray, starts: Sequence[int], win: int, step: int, hamming: bool = False, average: bool = True) -> np.ndarray:
    """`aggregate` for equal-sized windows held class-major: cm [C, windows, frames] -> [n_out, C].  The class axis is
    outermost so every slice added is contiguous (reductions and slices along a trailing axis of 3 or 7 entries are what
    numpy is slowest at)."""
    C, W, F = cm.shape
    n_out = n_frames_out(W, win, step)
    wgt = (np.hamming(F) if hamming else np.ones(F)).astype(np.float32)
    acc = np.zeros((C, n_out), dtype=np.float32)
    cnt = np.zeros(n_out, dtype=np.float32)
    cmw = cm.astype(np.float32, copy=False) * wgt if hamming else cm.astype(np.float32, copy=False)
    for w, s0 in enumerate(starts):
        f0 = int(np.rint(s0 / 270.0))
        acc[:, f0:f0 + F] += cmw[:, w]
        cnt[f0:f0 + F] += wgt
    if average:
        acc = acc / np.maximum(cnt, np.float32(1e-12))
    return np.ascontiguousarray(acc.T)


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
    regions = [(s, e) for s, e in regions if e - s > 1e-6]            # an empty segment never enters an Annotation
    merged: List[Tuple[float, float]] = []
    for s, e in regions:
        if min_off > 0.0 and merged and s - merged[-1][1] < min_off:   # Timeline.support(collar): gaps shorter than the collar
            merged[-1] = (merged[-1][0], e)
        else:
            merged.append((s, e))
    return [(s, e) for s, e in merged if e - s >= min_on]


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

    def _score(self, outs, starts):
        """Pre-aggregation hook (max over the speakers) + Hamming-weighted overlap-add average (Inference.slide)."""
        outs = np.stack(outs)                   # equal windows: class-major, one contiguous add per window
        if self.net.powerset:                   # a powerset model is decoded to hard multi-label first (Inference's conversion)
            sc = multilabel_cm(outs, True).max(axis=0)
        else:
            sc = outs[..., 0]
            for c in range(1, outs.shape[-1]):
                sc = np.maximum(sc, outs[..., c])
        return aggregate_cm(sc[None], starts, self.win, self.step, hamming=True)[:, 0]

    def begin(self, items: Sequence):
        """Cut the windows of all items and QUEUE the network; returns a handle for `finish`.  Nothing waits for the GPU
        here, so a caller can queue more work (e.g. the diarization network) before collecting."""
        xs = [load_mono_16k(it, self.net.device, getattr(self.net, "ctx", None)) for it in items]
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
            score = self._score(outs[i0:i0 + n], starts)
            n_valid = min(len(score), int(np.floor((len(x) / SR) / FRAME_STEP)) + 1)   # crop(Segment(0, duration), mode="loose")
            on, off = (0.5, 0.5) if self.net.powerset else (self.onset, self.offset)
            anns.append(Annotation([(s, e, "SPEECH") for s, e in binarize(score[:n_valid], on, off, self.min_on, self.min_off)]))
        return anns

    def batch(self, items: Sequence) -> List[Annotation]:
        """Several files / waveforms at once: all windows of all items go through the network together."""
        return self.finish(self.begin(items))

    def __call__(self, path_or_wave) -> Annotation:
        return self.batch([path_or_wave])[0]


def _large_clusters(lab: np.ndarray, min_size: int):
    ids, sizes = np.unique(lab, return_counts=True)
    return ids, sizes, ids[sizes >= min_size]


def agglomerative_centroid(emb: np.ndarray, threshold: float, min_cluster_size: int, min_clusters: int, max_clusters: int,
                           num_clusters: Optional[int] = None) -> np.ndarray:
    """pyannote AgglomerativeClustering.cluster (centroid linkage on unit-normalised embeddings, euclidean): cut the dendrogram at
    `threshold`; clusters of >= min(min_cluster_size, max(1, round(n / 10))) members are "large".  When the number of large
    clusters falls outside [min_clusters, max_clusters] (or `num_clusters` is given) the cut moves to the merge, closest in
    distance to the threshold, that yields the wanted number of large clusters.  Small clusters join the large cluster with the
    nearest centroid (cosine); labels are renumbered 0 .. k-1 in the order of scipy's flat-cluster ids."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import cdist
    n = emb.shape[0]
    if n == 1:
        return np.zeros(1, dtype=np.int64)
    big = min(min_cluster_size, max(1, round(0.1 * n)))
    e = np.asarray(emb, dtype=np.float64)
    e = e / np.linalg.norm(e, axis=-1, keepdims=True)
    Z = linkage(e, method="centroid", metric="euclidean")
    lab = fcluster(Z, threshold, criterion="distance") - 1
    ids, sizes, large = _large_clusters(lab, big)
    if len(large) < min_clusters:
        num_clusters = min_clusters
    elif len(large) > max_clusters:
        num_clusters = max_clusters
    if num_clusters is not None:
        Zi = Z.copy()
        Zi[:, 2] = np.arange(n - 1)                     # cut by merge index instead of by distance
        best_it, best_n = n - 1, 1                      # upstream's start: the last merge / one large cluster (the fallback when no cut fits)
        for it in np.argsort(np.abs(Z[:, 2] - threshold)):
            if Zi[it, 3] < big:                          # this merge cannot have changed the number of large clusters
                continue
            lab = fcluster(Zi, it, criterion="distance") - 1
            ids, sizes, large = _large_clusters(lab, big)
            if abs(len(large) - num_clusters) < abs(best_n - num_clusters):
                best_it, best_n = it, len(large)
            if len(large) == num_clusters:
                break
        if best_n != num_clusters:
            lab = fcluster(Zi, best_it, criterion="distance") - 1
            ids, sizes, large = _large_clusters(lab, big)
    if len(large) == 0:
        return np.zeros(n, dtype=np.int64)
    small = ids[sizes < big]
    if len(small):
        cl = np.vstack([e[lab == k].mean(axis=0) for k in large])
        cs = np.vstack([e[lab == k].mean(axis=0) for k in small])
        for j, k in enumerate(np.argmin(cdist(cl, cs, metric="cosine"), axis=0)):
            lab[lab == small[j]] = large[k]
        lab = np.unique(lab, return_inverse=True)[1]
    return lab.astype(np.int64)


def assign_to_centroids(embs: np.ndarray, train_rows: np.ndarray, train_labels: np.ndarray) -> np.ndarray:
    """BaseClustering.assign_embeddings (unconstrained): centroid k = mean of the training embeddings of cluster k; every
    embedding -- training or not -- goes to the centroid with the largest cosine similarity (a NaN row goes to 0)."""
    from scipy.spatial.distance import cdist
    e = np.asarray(embs, dtype=np.float64)
    cents = np.vstack([e[train_rows][train_labels == k].mean(axis=0) for k in range(int(train_labels.max()) + 1)])
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.argmax(2.0 - cdist(e, cents, metric="cosine"), axis=1).astype(np.int64)


class SpeakerDiarization:
    """pyannote/speaker-diarization-3.1: powerset segmentation over 10 s windows (10 % step) decoded to hard multi-label, one
    embedding per ACTIVE (window, local speaker) pooled over its overlap-free frames when more than `min_num_frames` remain,
    agglomerative clustering of the finite embeddings + nearest-centroid assignment of all of them, per-frame speaker count
    (rounded overlap-add average) x summed clustered activations -> top-`count` clusters per frame -> timeline.
    `min_num_frames` = ceil(589 * min_num_samples / 160000) with the embedding model's shortest usable input (one 25 ms
    fbank frame for the ResNet-34, recalled) = 2."""

    def __init__(self, seg_net, embedder, duration: float = 10.0, step_ratio: float = 0.1, threshold: float = 0.7045654963945799,
                 min_cluster_size: int = 12, min_duration_off: float = 0.0, min_num_frames: int = 2, batch: int = 32):
        self.net, self.emb = seg_net, embedder
        self.win, self.step = int(duration * SR), int(duration * step_ratio * SR)
        self.threshold, self.min_cluster_size, self.min_off, self.min_num_frames, self.batch_size = threshold, min_cluster_size, min_duration_off, min_num_frames, batch

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
        xs = [load_mono_16k(it, self.net.device, getattr(self.net, "ctx", None)) for it in items]
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
            cleanf = mc * (n_spk < 2)                                        # overlap-free activity
            n_act, n_clean = mc.sum(axis=-1), cleanf.sum(axis=-1)            # [speakers, windows] frame counts (exact in fp32)
            use_clean = n_clean > self.min_num_frames                        # prefer overlap-free frames when enough remain
            kw, ks = np.nonzero((n_act > 0).T)                               # every ACTIVE local speaker; row-major: window, then speaker
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
        lo = num_speakers or min_speakers or 1                               # set_num_speakers
        hi = num_speakers or max_speakers or None
        xs, per_item, embs = handle
        embs = np.zeros((0, getattr(self.emb, "DIM", 512)), dtype=np.float32) if embs is None else embs.cpu().numpy()
        anns, e0 = [], 0
        for x, it in zip(xs, per_item):
            if it is None or not it[2]:
                anns.append(Annotation([]))
                continue
            starts, mc, keys, n_spk = it
            anns.append(self._reconstruct(starts, mc, keys, embs[e0:e0 + len(keys)], num_speakers, lo, hi, n_spk))
            e0 += len(keys)
        return anns

    def _cluster(self, embs: np.ndarray, num: Optional[int], lo: int, hi: Optional[int]) -> np.ndarray:
        """BaseClustering.__call__ over the embeddings of the active local speakers: cluster the finite ones, assign all."""
        ok = np.flatnonzero(np.isfinite(embs).all(axis=1))
        n = len(ok)
        lo_c = max(1, min(n, num or lo or 1))                                # set_num_clusters
        hi_c = max(1, min(n, num or hi or n))
        if lo_c > hi_c:
            raise ValueError("min_speakers must not exceed max_speakers")
        if lo_c == hi_c:
            num = lo_c
        if hi_c < 2:
            return np.zeros(len(embs), dtype=np.int64)
        train = agglomerative_centroid(embs[ok], self.threshold, self.min_cluster_size, lo_c, hi_c, num)
        return assign_to_centroids(embs, ok, train)

    def _reconstruct(self, starts, mc, keys, embs, num: Optional[int], lo: int, hi: Optional[int], n_spk=None) -> Annotation:
        """mc: hard local activity [speakers, windows, frames]; keys: the ACTIVE (window, local speaker) pairs, embs their embeddings."""
        if n_spk is None:
            n_spk = mc.sum(axis=0)
        count = np.rint(aggregate_cm(n_spk[None], starts, self.win, self.step)[:, 0]).astype(np.int64)      # speaker_count
        if count.max() == 0:
            return Annotation([])
        labels = self._cluster(embs, num, lo, hi)
        if hi is not None:
            count = np.minimum(count, hi)
        n_clusters = int(labels.max()) + 1
        clustered = np.zeros((max(n_clusters, int(count.max())),) + mc.shape[1:], dtype=np.float32)
        for (kc, sp), lab in zip(keys, labels):                            # max over the local speakers of a cluster, per window
            np.maximum(clustered[lab, kc], mc[sp, kc], out=clustered[lab, kc])
        act = aggregate_cm(clustered, starts, self.win, self.step, average=False)                           # [frames, clusters]
        # to_diarization: at each frame the `count` most active clusters speak (stable order on ties)
        ranks = np.argsort(np.argsort(-act, axis=1, kind="stable"), axis=1, kind="stable")
        binary = (ranks < count[:, None]).astype(np.float32)
        tracks = []
        for c in range(binary.shape[1]):
            tracks += [(s, e, c) for s, e in binarize(binary[:, c], 0.5, 0.5, 0.0, self.min_off)]
        present = sorted({c for _, _, c in tracks})                        # labels() in sorted order take SPEAKER_00, SPEAKER_01, ...
        rename = {c: f"SPEAKER_{i:02d}" for i, c in enumerate(present)}
        return Annotation([(s, e, rename[c]) for s, e, c in tracks])
