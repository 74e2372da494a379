"""Interval glue of the hot path (reference back/api.py:294-343): merge same-speaker diarization
turns, clip a turn to the VAD speech hull, sweep-line overlap detection.  Pure Python on tiny
lists; behaviour is pinned by tests/golden/glue_intervals.json (reference-generated)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

Turn = Tuple[float, float, str]


def merge_diarization_segments(segments: Sequence[Turn], gap_threshold: float) -> List[Turn]:
    """Chronological merge of consecutive turns of the SAME label separated by <= gap_threshold.
    (The reference sorts its argument in place; callers here pass throw-away lists, so a sorted
    copy is equivalent.)"""
    turns = sorted(segments, key=lambda t: t[0])
    out: List[Turn] = []
    for s, e, lab in turns:
        if out and out[-1][2] == lab and (s - out[-1][1]) <= gap_threshold:
            out[-1] = (out[-1][0], e, lab)      # note: end is overwritten, not max()-ed, like the reference
        else:
            out.append((s, e, lab))
    return out


def get_vad_intervals(vad_annotation) -> List[Tuple[float, float]]:
    return [(seg.start, seg.end) for seg, _, _ in vad_annotation.itertracks(yield_label=True)]


def refine_segment_with_vad(segment: Tuple[float, float], vad_intervals: Sequence[Tuple[float, float]]) -> Optional[Tuple[float, float]]:
    """Hull of the intersections of `segment` with the speech intervals, or None if disjoint."""
    s0, e0 = segment
    lo, hi = None, None
    for vs, ve in vad_intervals:
        a, b = max(s0, vs), min(e0, ve)
        if a < b:
            lo = a if lo is None or a < lo else lo
            hi = b if hi is None or b > hi else hi
    return None if lo is None else (lo, hi)


def find_segment_overlaps(segments: Sequence[Turn]) -> Dict[Tuple[float, float], List[str]]:
    """Sweep line over turn boundaries (ends before starts at equal times).  A region is emitted at
    every END event reached while more than one label is active, spanning from the moment the
    active set first exceeded one label; later regions with the same (start, end) key overwrite
    earlier ones.  Speaker lists are returned sorted (the reference's order comes from a set)."""
    events = []
    for s, e, lab in segments:
        events.append((s, 1, lab))
        events.append((e, -1, lab))
    events.sort(key=lambda ev: (ev[0], ev[1]))
    active = set()
    since: Optional[float] = None
    found: Dict[Tuple[float, float], List[str]] = {}
    for t, kind, lab in events:
        if kind == 1:
            active.add(lab)
            if len(active) > 1 and since is None:
                since = t
        else:
            if len(active) > 1 and since is not None:
                found[(since, t)] = sorted(active)
            active.discard(lab)
            if len(active) <= 1:
                since = None
    return found
