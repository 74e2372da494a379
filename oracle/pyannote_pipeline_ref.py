"""CPU restatement (plain Python loops over numpy) of the HOST side of the two pyannote pipelines the reference calls:
`pyannote/voice-activity-detection` (/root/reference/back/api.py:782-786, called at 1311-1312) and
`pyannote/speaker-diarization-3.1` (back/api.py:788-792, called at 1052-1064 and 1120-1137).

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline).  The product's version is clearconverse_amd/pipelines.py
(vectorised numpy); this file is written frame by frame so that the two share nothing but numpy and scipy.

Restates, from recollection of the published pyannote.audio 3.1 sources [UPSTREAM-RECALL] (un-pinned dependency,
back/requirements.txt:12-19, not vendored, not installed here; pyannote.core for the Annotation side):
  core/inference.py::Inference.slide        chunk layout (complete chunks every `step`, one zero-padded last chunk)
  core/inference.py::Inference.aggregate    overlap-add of per-chunk frame scores (Hamming weights or flat, average or sum)
  utils/powerset.py::Powerset.to_multilabel argmax class -> its speaker set (class order: combinations by set size)
  utils/signal.py::Binarize.__call__        hysteresis thresholding, Timeline.support(collar), min_duration_on
  pipelines/voice_activity_detection.py::VoiceActivityDetection.apply
  pipelines/utils/diarization.py::SpeakerDiarizationMixin.speaker_count / to_diarization / to_annotation / set_num_speakers
  pipelines/speaker_diarization.py::SpeakerDiarization.get_embeddings / reconstruct / apply
  pipelines/clustering.py::BaseClustering.filter_embeddings / set_num_clusters / assign_embeddings / __call__ and
                           AgglomerativeClustering.cluster   (scipy linkage / fcluster / cdist, which upstream calls too)

PARITY STATUS: **parity unpinned** -- the reference holds no fixture for these pipelines and pyannote.audio is not importable
here; a misremembered detail of upstream is invisible to every test in this repo.  Decisions taken from recollection are
marked (?) where they are least certain.
"""
from __future__ import annotations

import itertools
import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

SR = 16000
RF_STEP = 270           # SincNet receptive-field step in samples (stride 10 x three MaxPool1d(3))
RF_SIZE = 991           # SincNet receptive-field size in samples
FRAME_STEP = RF_STEP / SR
FRAME_DUR = RF_SIZE / SR


# ----------------------------------------------------------------------------------------------- sliding inference
def chunk_starts(num_samples: int, window: int, step: int) -> Tuple[List[int], bool]:
    """Inference.slide: `waveform.unfold(1, window, step)` complete chunks, plus one last chunk (zero padded to `window`)
    when the signal is shorter than a window or the complete chunks do not end exactly at the last sample."""
    starts: List[int] = []
    num_chunks = 0
    if num_samples >= window:
        num_chunks = (num_samples - window) // step + 1
        for i in range(num_chunks):
            starts.append(i * step)
    has_last = num_samples < window or (num_samples - window) % step > 0
    if has_last:
        starts.append(num_chunks * step)
    return starts, has_last


def cut_chunks(wave: np.ndarray, window: int, step: int) -> Tuple[List[int], List[np.ndarray]]:
    starts, _ = chunk_starts(len(wave), window, step)
    chunks = []
    for s in starts:
        c = np.zeros(window, dtype=np.float32)
        part = wave[s:s + window]
        c[: len(part)] = part
        chunks.append(c)
    return starts, chunks


# ----------------------------------------------------------------------------------------------- powerset
def powerset_mapping(num_classes: int = 3, max_set_size: int = 2) -> List[List[int]]:
    """Powerset.build_mapping: row k = the speaker set of powerset class k (empty set, singles, pairs ...)."""
    rows = []
    for size in range(0, max_set_size + 1):
        for comb in itertools.combinations(range(num_classes), size):
            rows.append([1 if c in comb else 0 for c in range(num_classes)])
    return rows


def powerset_to_multilabel(logp: np.ndarray, num_classes: int = 3, max_set_size: int = 2) -> np.ndarray:
    """Powerset.to_multilabel(soft=False): one-hot of the argmax class times the mapping.  [frames, 7] -> [frames, 3]."""
    mapping = powerset_mapping(num_classes, max_set_size)
    out = np.zeros((logp.shape[0], num_classes), dtype=np.float32)
    for t in range(logp.shape[0]):
        best, best_v = 0, logp[t, 0]
        for k in range(1, logp.shape[1]):
            if logp[t, k] > best_v:                 # first maximum wins a tie (argmax)
                best, best_v = k, logp[t, k]
        for c in range(num_classes):
            out[t, c] = mapping[best][c]
    return out


# ----------------------------------------------------------------------------------------------- aggregation
def closest_frame(t: float) -> int:
    """SlidingWindow(start=0, duration=FRAME_DUR, step=FRAME_STEP).closest_frame(t)."""
    return int(np.rint((t - 0.5 * FRAME_DUR) / FRAME_STEP))


def aggregate(scores: Sequence[np.ndarray], starts: Sequence[int], window: int, step: int, hamming: bool, skip_average: bool,
              missing: float = 0.0, epsilon: float = 1e-12) -> np.ndarray:
    """Inference.aggregate (warm_up = (0, 0)): scores[c] is [frames_per_chunk, classes] (NaN = no prediction).  float32
    accumulators as upstream.  Returns [num_frames, classes] on the frame grid that starts at the first chunk."""
    num_chunks = len(scores)
    fpc, num_classes = scores[0].shape
    ham = np.hamming(fpc) if hamming else np.ones(fpc)
    chunk_dur, chunk_step = window / SR, step / SR
    num_frames = closest_frame(0.0 + chunk_dur + (num_chunks - 1) * chunk_step + 0.5 * FRAME_DUR) + 1
    out = np.zeros((num_frames, num_classes), dtype=np.float32)
    cnt = np.zeros((num_frames, num_classes), dtype=np.float32)
    seen = np.zeros((num_frames, num_classes), dtype=np.float32)
    for c in range(num_chunks):
        start_frame = closest_frame(starts[c] / SR + 0.5 * FRAME_DUR)
        for f in range(fpc):
            w = np.float32(ham[f])
            for k in range(num_classes):
                v = scores[c][f, k]
                m = np.float32(0.0) if np.isnan(v) else np.float32(1.0)
                v = np.float32(0.0) if np.isnan(v) else np.float32(v)
                out[start_frame + f, k] += v * m * w
                cnt[start_frame + f, k] += m * w
                seen[start_frame + f, k] = max(seen[start_frame + f, k], m)
    if not skip_average:
        out = out / np.maximum(cnt, np.float32(epsilon))
    out[seen == 0.0] = missing
    return out


def crop_loose(num_frames: int, num_samples: int) -> int:
    """SlidingWindowFeature.crop(Segment(0, duration), mode="loose"): frames 0 .. floor(duration / step) that exist."""
    j = int(math.floor((num_samples / SR) / FRAME_STEP))
    return min(num_frames, j + 1)


# ----------------------------------------------------------------------------------------------- binarize
def binarize(scores: np.ndarray, onset: float, offset: float, min_duration_on: float = 0.0, min_duration_off: float = 0.0
             ) -> List[Tuple[float, float, int]]:
    """Binarize.__call__ on a [frames, classes] feature whose frame i is centred at i * FRAME_STEP + FRAME_DUR / 2
    (pad_onset = pad_offset = 0).  Returns (start, end, class) regions sorted by (start, end)."""
    num_frames, num_classes = scores.shape
    ts = [i * FRAME_STEP + 0.5 * FRAME_DUR for i in range(num_frames)]
    out: List[Tuple[float, float, int]] = []
    for k in range(num_classes):
        regions: List[List[float]] = []
        if num_frames == 0:
            continue
        start = ts[0]
        active = bool(scores[0, k] > onset)
        t = ts[0]
        for i in range(1, num_frames):
            t, y = ts[i], scores[i, k]
            if active:
                if y < offset:
                    regions.append([start, t])
                    start, active = t, False
            elif y > onset:
                start, active = t, True
        if active:
            regions.append([start, t])
        regions = [r for r in regions if r[1] - r[0] > 1e-6]       # Annotation.__setitem__ ignores empty segments
        if min_duration_off > 0.0:
            # Annotation.support(collar): per label, merge segments whose gap is shorter than the collar (or absent)
            merged: List[List[float]] = []
            for s, e in sorted(regions):
                if merged and (s <= merged[-1][1] or s - merged[-1][1] < min_duration_off):
                    merged[-1][1] = max(merged[-1][1], e)
                else:
                    merged.append([s, e])
            regions = merged
        if min_duration_on > 0.0:
            regions = [r for r in regions if r[1] - r[0] >= min_duration_on]
        out += [(s, e, k) for s, e in regions]
    return sorted(out, key=lambda r: (r[0], r[1], r[2]))


# ----------------------------------------------------------------------------------------------- VAD pipeline
def voice_activity_detection(wave: np.ndarray, seg_fn: Callable[[List[np.ndarray]], List[np.ndarray]], powerset: bool,
                             duration: float = 5.0, step_ratio: float = 0.1, onset: float = 0.767, offset: float = 0.377,
                             min_duration_on: float = 0.136, min_duration_off: float = 0.067) -> List[Tuple[float, float]]:
    """VoiceActivityDetection.apply: sliding segmentation, pre-aggregation hook max over the speakers, Hamming-weighted
    overlap-add average (Inference.slide aggregates with hamming=True, missing=0), crop to the file, Binarize.
    `seg_fn(chunks)` returns one [frames, classes] score array per chunk: sigmoid multi-label scores, or powerset
    log-probabilities (converted to hard multi-label first, with onset = offset = 0.5, as upstream does for powerset models)."""
    window, step = int(duration * SR), int(round(duration * step_ratio * SR))
    starts, chunks = cut_chunks(wave, window, step)
    raw = seg_fn(chunks)
    if powerset:
        raw = [powerset_to_multilabel(r) for r in raw]
        onset = offset = 0.5
    hooked = []
    for r in raw:
        m = np.zeros((r.shape[0], 1), dtype=np.float32)
        for f in range(r.shape[0]):
            m[f, 0] = max(r[f, k] for k in range(r.shape[1]))
        hooked.append(m)
    agg = aggregate(hooked, starts, window, step, hamming=True, skip_average=False, missing=0.0)
    agg = agg[: crop_loose(agg.shape[0], len(wave))]
    return [(s, e) for s, e, _ in binarize(agg, onset, offset, min_duration_on, min_duration_off)]


# ----------------------------------------------------------------------------------------------- clustering
def set_num_clusters(num_embeddings: int, num_clusters, min_clusters, max_clusters):
    """BaseClustering.set_num_clusters."""
    min_clusters = num_clusters or min_clusters or 1
    min_clusters = max(1, min(num_embeddings, min_clusters))
    max_clusters = num_clusters or max_clusters or num_embeddings
    max_clusters = max(1, min(num_embeddings, max_clusters))
    if min_clusters > max_clusters:
        raise ValueError("min_clusters must be smaller than (or equal to) max_clusters")
    if min_clusters == max_clusters:
        num_clusters = min_clusters
    return num_clusters, min_clusters, max_clusters


def agglomerative_cluster(embeddings: np.ndarray, threshold: float, min_cluster_size: int, min_clusters: int, max_clusters: int,
                          num_clusters: Optional[int] = None) -> np.ndarray:
    """AgglomerativeClustering.cluster (method="centroid", metric="cosine": unit-normalise, euclidean centroid linkage)."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import cdist
    embeddings = np.array(embeddings, dtype=np.float64)
    num_embeddings = embeddings.shape[0]
    min_cluster_size = min(min_cluster_size, max(1, round(0.1 * num_embeddings)))
    if num_embeddings == 1:
        return np.zeros((1,), dtype=np.int64)
    embeddings /= np.linalg.norm(embeddings, axis=-1, keepdims=True)
    dendrogram = linkage(embeddings, method="centroid", metric="euclidean")
    clusters = fcluster(dendrogram, threshold, criterion="distance") - 1
    uniq, counts = np.unique(clusters, return_counts=True)
    large = uniq[counts >= min_cluster_size]
    num_large = len(large)
    if num_large < min_clusters:
        num_clusters = min_clusters
    elif num_large > max_clusters:
        num_clusters = max_clusters
    if num_clusters is not None:
        # stop by iteration index instead of by distance, going further and further away from the threshold
        _dend = np.copy(dendrogram)
        _dend[:, 2] = np.arange(num_embeddings - 1)
        # "best_iteration = num_embeddings - 1; best_num_large_clusters = 1": when no merge gives the wanted number of large clusters
        # the fallback is the candidate closest to it, and if none beats ONE cluster, the last merge (everything in one cluster)
        best_iteration = num_embeddings - 1
        best_num_large = 1
        for iteration in np.argsort(np.abs(dendrogram[:, 2] - threshold)):
            if _dend[iteration, 3] < min_cluster_size:
                continue
            clusters = fcluster(_dend, iteration, criterion="distance") - 1
            uniq, counts = np.unique(clusters, return_counts=True)
            large = uniq[counts >= min_cluster_size]
            num_large = len(large)
            if abs(num_large - num_clusters) < abs(best_num_large - num_clusters):
                best_iteration, best_num_large = iteration, num_large
            if num_large == num_clusters:
                break
        if best_num_large != num_clusters:
            clusters = fcluster(_dend, best_iteration, criterion="distance") - 1
            uniq, counts = np.unique(clusters, return_counts=True)
            large = uniq[counts >= min_cluster_size]
            num_large = len(large)
    if num_large == 0:
        clusters[:] = 0
        return clusters.astype(np.int64)
    small = uniq[counts < min_cluster_size]
    if len(small) == 0:
        return clusters.astype(np.int64)
    large_centroids = np.vstack([np.mean(embeddings[clusters == k], axis=0) for k in large])
    small_centroids = np.vstack([np.mean(embeddings[clusters == k], axis=0) for k in small])
    d = cdist(large_centroids, small_centroids, metric="cosine")
    for small_k, large_k in enumerate(np.argmin(d, axis=0)):
        clusters[clusters == small[small_k]] = large[large_k]
    _, clusters = np.unique(clusters, return_inverse=True)
    return clusters.astype(np.int64)


def clustering(embeddings: np.ndarray, segmentations: np.ndarray, threshold: float, min_cluster_size: int, num_clusters, min_clusters,
               max_clusters) -> np.ndarray:
    """BaseClustering.__call__: embeddings [chunks, speakers, dim], segmentations [chunks, frames, speakers] (binary).
    Train on the embeddings of ACTIVE local speakers without NaN (filter_embeddings), then assign EVERY (chunk, speaker)
    to the centroid with the largest cosine similarity (assign_embeddings, constrained=False).  -> hard clusters [chunks, speakers]."""
    from scipy.spatial.distance import cdist
    num_chunks, num_speakers, _ = embeddings.shape
    train_idx = []
    for c in range(num_chunks):
        for s in range(num_speakers):
            active = float(np.sum(segmentations[c][:, s])) > 0
            valid = not bool(np.any(np.isnan(embeddings[c, s])))
            if active and valid:
                train_idx.append((c, s))
    train = np.stack([embeddings[c, s] for c, s in train_idx]) if train_idx else np.zeros((0, embeddings.shape[2]))
    num_clusters, min_clusters, max_clusters = set_num_clusters(len(train_idx), num_clusters, min_clusters, max_clusters)
    if max_clusters < 2:
        return np.zeros((num_chunks, num_speakers), dtype=np.int64)
    train_clusters = agglomerative_cluster(train, threshold, min_cluster_size, min_clusters, max_clusters, num_clusters)
    k_max = int(np.max(train_clusters)) + 1
    train64 = np.asarray(train, dtype=np.float64)
    centroids = np.vstack([np.mean(train64[train_clusters == k], axis=0) for k in range(k_max)])
    flat = np.asarray(embeddings, dtype=np.float64).reshape(num_chunks * num_speakers, -1)
    with np.errstate(invalid="ignore", divide="ignore"):
        soft = 2.0 - cdist(flat, centroids, metric="cosine")
    return np.argmax(soft.reshape(num_chunks, num_speakers, k_max), axis=2).astype(np.int64)


# ----------------------------------------------------------------------------------------------- diarization pipeline
def speaker_count(segmentations: Sequence[np.ndarray], starts, window, step) -> np.ndarray:
    """SpeakerDiarizationMixin.speaker_count (warm_up (0, 0)): rint of the flat overlap-add AVERAGE of the per-frame number
    of active local speakers.  -> [num_frames] integers."""
    sums = []
    for seg in segmentations:
        s = np.zeros((seg.shape[0], 1), dtype=np.float32)
        for f in range(seg.shape[0]):
            s[f, 0] = sum(float(seg[f, k]) for k in range(seg.shape[1]))
        sums.append(s)
    avg = aggregate(sums, starts, window, step, hamming=False, skip_average=False, missing=0.0)
    return np.rint(avg[:, 0]).astype(np.int64)


def embedding_masks(seg: np.ndarray, min_num_frames: int) -> List[np.ndarray]:
    """SpeakerDiarization.get_embeddings, exclude_overlap=True: per local speaker the overlap-free frames when MORE than
    `min_num_frames` of them remain, the whole activity otherwise."""
    frames, speakers = seg.shape
    clean = np.zeros_like(seg)
    for f in range(frames):
        n = sum(float(seg[f, k]) for k in range(speakers))
        for k in range(speakers):
            clean[f, k] = seg[f, k] if n < 2 else 0.0
    out = []
    for k in range(speakers):
        out.append(clean[:, k].copy() if float(np.sum(clean[:, k])) > min_num_frames else seg[:, k].copy())
    return out


def reconstruct(segmentations: Sequence[np.ndarray], hard_clusters: np.ndarray, count: np.ndarray, starts, window, step) -> np.ndarray:
    """SpeakerDiarization.reconstruct + to_diarization: per chunk the max over the local speakers of each cluster (NaN where
    a cluster is absent from the chunk), flat overlap-add SUM, then at every frame the `count` most active clusters speak."""
    num_clusters = int(np.max(hard_clusters)) + 1
    clustered = []
    for c, seg in enumerate(segmentations):
        cs = np.full((seg.shape[0], num_clusters), np.nan, dtype=np.float32)
        for k in sorted(set(int(x) for x in hard_clusters[c])):
            if k == -2:
                continue
            for f in range(seg.shape[0]):
                cs[f, k] = max(float(seg[f, s]) for s in range(seg.shape[1]) if hard_clusters[c, s] == k)
        clustered.append(cs)
    act = aggregate(clustered, starts, window, step, hamming=False, skip_average=True, missing=0.0)
    max_per_frame = int(np.max(count))
    if act.shape[1] < max_per_frame:
        act = np.pad(act, ((0, 0), (0, max_per_frame - act.shape[1])))
    n = min(act.shape[0], count.shape[0])
    act, count = act[:n], count[:n]
    binary = np.zeros_like(act)
    for t in range(n):
        order = np.argsort(-act[t], kind="stable")
        for i in range(int(count[t])):
            binary[t, order[i]] = 1.0
    return binary


def speaker_diarization(wave: np.ndarray, seg_fn: Callable[[List[np.ndarray]], List[np.ndarray]],
                        emb_fn: Callable[[np.ndarray, np.ndarray], np.ndarray], num_speakers=None, min_speakers=None,
                        max_speakers=None, duration: float = 10.0, step_ratio: float = 0.1, threshold: float = 0.7045654963945799,
                        min_cluster_size: int = 12, min_duration_off: float = 0.0, min_num_frames: int = 2,
                        return_internals: bool = False):
    """SpeakerDiarization.apply for a powerset segmentation model.  `seg_fn(chunks)` -> per chunk [frames, 7] log-probabilities;
    `emb_fn(chunk_waveform, frame_mask)` -> the embedding of one (chunk, local speaker) (NaN when it cannot be computed).
    `min_num_frames` = ceil(frames_per_chunk * min_num_samples / chunk_samples) with the embedding model's min_num_samples
    (400 samples = one 25 ms fbank frame for the WeSpeaker ResNet-34 (?) -> ceil(589 * 400 / 160000) = 2).
    Returns [(start, end, "SPEAKER_xx")] sorted by start."""
    # set_num_speakers
    min_speakers = num_speakers or min_speakers or 1
    max_speakers = num_speakers or max_speakers or np.inf
    window, step = int(duration * SR), int(round(duration * step_ratio * SR))
    starts, chunks = cut_chunks(wave, window, step)
    segs = [powerset_to_multilabel(r) for r in seg_fn(chunks)]         # hard multi-label [frames, 3] per chunk
    count = speaker_count(segs, starts, window, step)
    if int(np.max(count)) == 0:
        return ([], {}) if return_internals else []
    embs = []
    for c, seg in enumerate(segs):
        embs.append(np.stack([np.asarray(emb_fn(chunks[c], m), dtype=np.float32) for m in embedding_masks(seg, min_num_frames)]))
    embeddings = np.stack(embs)                                         # [chunks, speakers, dim]
    hard = clustering(embeddings, np.stack(segs), threshold, min_cluster_size, num_speakers, min_speakers,
                      None if max_speakers == np.inf else max_speakers)
    count = np.minimum(count, max_speakers).astype(np.int64)
    for c, seg in enumerate(segs):                                      # inactive local speakers take no part in the reconstruction
        for s in range(seg.shape[1]):
            if float(np.sum(seg[:, s])) == 0:
                hard[c, s] = -2
    binary = reconstruct(segs, hard, count, starts, window, step)
    regions = binarize(binary, 0.5, 0.5, 0.0, min_duration_off)
    present = sorted({k for _, _, k in regions})
    name = {k: f"SPEAKER_{i:02d}" for i, k in enumerate(present)}       # labels() sorted, zipped with the class generator
    out = [(s, e, name[k]) for s, e, k in regions]
    if return_internals:
        return out, dict(starts=starts, segmentations=segs, count=count, hard_clusters=hard, embeddings=embeddings, binary=binary)
    return out
