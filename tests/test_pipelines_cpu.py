"""CPU: the host-side post-net of the VAD / diarization pipelines (numpy), on hand-checkable inputs."""
import numpy as np

from clearconverse_amd import pipelines as P


def test_sliding_chunks_cover_the_signal():
    assert P.sliding_chunks(100, 160, 16) == [0]
    s = P.sliding_chunks(480000, 160000, 16000)
    assert s[0] == 0 and s[-1] + 160000 >= 480000 and all(b - a == 16000 for a, b in zip(s, s[1:]))
    s = P.sliding_chunks(170001, 160000, 16000)
    assert s == [0, 16000]


def test_powerset_decoding():
    lp = np.full((4, 7), -10.0, dtype=np.float32)
    lp[0, 0] = 0; lp[1, 2] = 0; lp[2, 4] = 0; lp[3, 6] = 0      # empty, {1}, {0,1}, {1,2}
    m = P.powerset_to_multilabel(lp)
    assert m.tolist() == [[0, 0, 0], [0, 1, 0], [1, 1, 0], [0, 1, 1]]


def test_aggregate_is_an_overlap_average():
    a = np.ones((10, 1), dtype=np.float32); b = 3 * np.ones((10, 1), dtype=np.float32)
    out = P.aggregate([a, b], [0, 5 * 270], 10 * 270, 5 * 270)
    assert out.shape[0] == P.n_frames_out(2, 10 * 270, 5 * 270) == 16
    assert np.allclose(out[:5, 0], 1) and np.allclose(out[5:10, 0], 2) and np.allclose(out[10:15, 0], 3) and out[15, 0] == 0
    tot = P.aggregate([a, b], [0, 5 * 270], 10 * 270, 5 * 270, average=False)
    assert np.allclose(tot[:5, 0], 1) and np.allclose(tot[5:10, 0], 4) and np.allclose(tot[10:15, 0], 3)
    ham = P.aggregate([a, b], [0, 5 * 270], 10 * 270, 5 * 270, hamming=True)          # Hamming-weighted average of 1 and 3
    w = np.hamming(10)
    assert np.allclose(ham[5:10, 0], (w[5:] * 1 + w[:5] * 3) / (w[5:] + w[:5]), atol=1e-6)


def test_binarize_hysteresis_and_duration_rules():
    y = np.array([0, .9, .6, .5, .3, .2, .9, .9, 0, 0, .9, 0], dtype=np.float32)
    r = P.binarize(y, onset=0.8, offset=0.4, frame_step=1.0, t0=0.0)
    assert r == [(1.0, 4.0), (6.0, 8.0), (10.0, 11.0)]
    r = P.binarize(y, 0.8, 0.4, min_on=1.5, min_off=0.0, frame_step=1.0, t0=0.0)
    assert r == [(1.0, 4.0), (6.0, 8.0)]
    r = P.binarize(y, 0.8, 0.4, min_on=0.0, min_off=2.5, frame_step=1.0, t0=0.0)
    assert r == [(1.0, 11.0)]


def test_agglomerative_clustering_respects_speaker_bounds():
    rng = np.random.default_rng(0)
    a = rng.normal(0, 0.05, (20, 16)) + np.eye(16)[0] * 3
    b = rng.normal(0, 0.05, (20, 16)) + np.eye(16)[1] * 3
    c = rng.normal(0, 0.05, (3, 16)) + np.eye(16)[2] * 3
    e = np.concatenate([a, b, c])
    lab = P.agglomerative_centroid(e, threshold=0.7, min_cluster_size=5, min_clusters=1, max_clusters=2)
    assert len(set(lab)) == 2 and len(set(lab[:20])) == 1 and len(set(lab[20:40])) == 1 and lab[0] != lab[20]
    lab3 = P.agglomerative_centroid(e, threshold=0.7, min_cluster_size=2, min_clusters=1, max_clusters=20)
    assert len(set(lab3)) == 3
    one = P.agglomerative_centroid(e, threshold=5.0, min_cluster_size=2, min_clusters=2, max_clusters=2)
    assert len(set(one)) == 2


def test_annotation_contract():
    ann = P.Annotation([(2.0, 3.0, "SPEAKER_01"), (0.5, 1.5, "SPEAKER_00")])
    got = [(seg.start, seg.end, lab) for seg, _, lab in ann.itertracks(yield_label=True)]
    assert got == [(0.5, 1.5, "SPEAKER_00"), (2.0, 3.0, "SPEAKER_01")]


def test_binarize_vectorised_state_machine_equals_the_loop():
    """The numpy forward-fill form of the hysteresis state machine against the frame-by-frame loop it replaces, on random
    scores (plateaus exactly at the thresholds included)."""
    from clearconverse_amd.pipelines import FRAME_DUR, FRAME_STEP, binarize

    def loop(score, onset, offset, min_on, min_off):
        times = 0.5 * FRAME_DUR + FRAME_STEP * np.arange(score.shape[0])
        regions, active, start = [], False, 0.0
        for t, y in zip(times, score):
            if active:
                if y < offset:
                    regions.append((start, float(t))); active = False
            elif y > onset:
                start, active = float(t), True
        if active:
            regions.append((start, float(times[-1])))
        regions = [(s, e) for s, e in regions if e - s > 1e-6]
        merged = []
        for s, e in regions:
            if min_off > 0.0 and merged and s - merged[-1][1] < min_off:
                merged[-1] = (merged[-1][0], e)
            else:
                merged.append((s, e))
        return [(s, e) for s, e in merged if e - s >= min_on]

    rng = np.random.default_rng(0)
    for trial in range(200):
        n = int(rng.integers(0, 400))
        score = np.round(rng.random(n), 1)                       # many exact ties with the thresholds
        onset = float(rng.choice([0.3, 0.5, 0.7, 0.767]))
        offset = float(rng.choice([0.2, 0.3, 0.377, onset]))
        if offset > onset:
            offset = onset
        min_on, min_off = float(rng.choice([0.0, 0.136])), float(rng.choice([0.0, 0.067]))
        assert binarize(score, onset, offset, min_on, min_off) == loop(score, onset, offset, min_on, min_off), trial
    assert binarize(np.zeros(0), 0.5, 0.5) == []
    assert binarize(np.array([0.9]), 0.5, 0.4) == []             # a single active frame has zero length


def test_host_postnet_matches_fixture():
    """VAD + diarization host logic over seeded fake networks (tests/fake_nets.py) against the committed fixture:
    window plans, powerset -> multi-label, pooling-mask choice, clustering, overlap-add, top-`count` selection and
    binarisation give the same timelines as when the fixture was written (a regression pin of the build's own
    post-net; the pyannote pipelines themselves are not in the image: parity unpinned)."""
    import json
    from tests import fake_nets
    got = json.loads(json.dumps(fake_nets.run()))
    with open(fake_nets.GOLDEN) as f:
        want = json.load(f)
    assert got == want


def test_aggregate_cm_equals_aggregate():
    rng = np.random.default_rng(5)
    cm = rng.random((3, 6, 40)).astype(np.float32)
    starts = [0, 2700, 5400, 8100, 10800, 13500]
    for kw in (dict(), dict(hamming=True), dict(average=False)):
        a = P.aggregate([np.ascontiguousarray(cm[:, w].T) for w in range(6)], starts, 40 * 270, 2700, **kw)
        b = P.aggregate_cm(cm, starts, 40 * 270, 2700, **kw)
        assert np.array_equal(a, b)


def test_multilabel_cm_equals_powerset_to_multilabel():
    rng = np.random.default_rng(6)
    lp = rng.standard_normal((4, 50, 7)).astype(np.float32)
    want = np.moveaxis(P.powerset_to_multilabel(lp), -1, 0)
    assert np.array_equal(P.multilabel_cm(lp, True), want)
    sc = rng.random((4, 50, 3)).astype(np.float32)
    assert np.array_equal(P.multilabel_cm(sc, False), np.moveaxis((sc > 0.5).astype(np.float32), -1, 0))


# ---------------------------------------------------------------------------------------------------------------------------
# A13 / A14 (reference back/api.py:1311-1312, 1052-1064, 1120-1137): the product's vectorised post-net against the frame-by-frame
# restatement of the pyannote pipelines in oracle/pyannote_pipeline_ref.py, on seeded fake-network scores.  Parity unpinned
# (the oracle itself is recalled, not pinned by a reference fixture) -- what this pins is that the two independent codings agree.
def _tracks(ann):
    return [(s.start, s.end, l) for s, _, l in ann.itertracks(yield_label=True)]


def test_pipeline_helpers_equal_the_oracle_loops():
    from oracle import pyannote_pipeline_ref as O
    rng = np.random.default_rng(11)
    for n in (1000, 79999, 80000, 80001, 88000, 480000, 473000, 160000, 171234):
        for win, step in ((80000, 8000), (160000, 16000)):
            assert P.sliding_chunks(n, win, step) == O.chunk_starts(n, win, step)[0], (n, win)
    lp = rng.standard_normal((300, 7)).astype(np.float32)
    lp[5] = 0.0                                                        # an exact tie: the first class wins in both
    assert np.array_equal(P.powerset_to_multilabel(lp), O.powerset_to_multilabel(lp))
    starts = [0, 8000, 16000, 24000, 32000]
    sc = [rng.random((293, 2)).astype(np.float32) for _ in starts]
    for hamming, average in ((True, True), (False, True), (False, False)):
        got = P.aggregate(sc, starts, 80000, 8000, hamming=hamming, average=average)
        want = O.aggregate(sc, starts, 80000, 8000, hamming=hamming, skip_average=not average)
        assert got.shape == want.shape and np.array_equal(got, want), (hamming, average)
    for trial in range(60):
        y = np.round(rng.random(int(rng.integers(1, 300))), 1).astype(np.float32)
        on = float(rng.choice([0.5, 0.767])); off = float(rng.choice([0.377, on])); mon = float(rng.choice([0.0, 0.136])); moff = float(rng.choice([0.0, 0.067]))
        assert P.binarize(y, on, off, mon, moff) == [(s, e) for s, e, _ in O.binarize(y[:, None], on, off, mon, moff)], trial


def test_clustering_equals_the_oracle():
    from oracle import pyannote_pipeline_ref as O
    rng = np.random.default_rng(3)
    for trial in range(40):
        k = int(rng.integers(1, 5))
        n = int(rng.integers(2, 70))
        cents = rng.standard_normal((k, 24)) * 2.0
        e = (cents[rng.integers(0, k, n)] + rng.standard_normal((n, 24)) * float(rng.choice([0.2, 0.8]))).astype(np.float32)
        lo, hi = [(1, 2), (1, 20), (2, 2), (1, 1), (3, 4)][trial % 5]
        lo_c, hi_c = max(1, min(n, lo)), max(1, min(n, hi))
        num = lo_c if lo_c == hi_c else None
        mcs = int(rng.choice([2, 12]))
        got = P.agglomerative_centroid(e, 0.7045654963945799, mcs, lo_c, hi_c, num)
        want = O.agglomerative_cluster(e, 0.7045654963945799, mcs, lo_c, hi_c, num)
        assert np.array_equal(got, want), trial


def test_vad_and_diarization_equal_the_oracle_pipelines_on_fake_nets():
    """Whole pipelines: VoiceActivityDetection / SpeakerDiarization (product) against oracle voice_activity_detection /
    speaker_diarization fed by the SAME seeded fake networks.  Identical labels, boundaries equal (same frame grid, float64)."""
    import torch
    from oracle import pyannote_pipeline_ref as O
    from tests import fake_nets
    for i, n in enumerate((480000, 473000, 163000, 90000, 52000)):
        wave = np.random.default_rng(100 + i).standard_normal(n).astype(np.float32)
        item = {"waveform": torch.from_numpy(wave), "sample_rate": 16000}
        # VAD: multi-label scores (the bench's model) and a powerset model used as VAD
        for powerset, ncls in ((False, 3), (True, 7)):
            net = fake_nets.FakeSeg(ncls, powerset, seed=20 + i)
            got = [(s.start, s.end) for s, _ in P.VoiceActivityDetection(net).batch([item])[0].itertracks()]
            want = O.voice_activity_detection(wave, lambda ch: net.segment_fetch([len(c) for c in ch]), powerset)
            assert got == want, (i, powerset)
        for kw in (dict(min_speakers=1, max_speakers=2), dict(), dict(num_speakers=3), dict(min_speakers=2, max_speakers=4), dict(max_speakers=1)):
            net = fake_nets.FakeSeg(7, True, seed=40 + i)
            emb = fake_nets.ContentEmb(n_voices=3 + i % 2)
            got = _tracks(P.SpeakerDiarization(net, emb).batch([item], **kw)[0])
            want = O.speaker_diarization(wave, lambda ch: net.segment_fetch([len(c) for c in ch]), emb.emb_fn, **kw)
            want = sorted(want, key=lambda t: (t[0], t[1]))
            assert [l for _, _, l in got] == [l for _, _, l in want], (i, kw)
            assert [(s, e) for s, e, _ in got] == [(s, e) for s, e, _ in want], (i, kw)
            assert len(got) > 0


def test_clustering_fallback_when_no_cut_gives_the_wanted_number_of_large_clusters():
    """pyannote's AgglomerativeClustering.cluster starts its search from (best_iteration = n - 1, best_num_large_clusters = 1)
    [UPSTREAM-RECALL, parity unpinned].  When the wanted number of clusters exceeds the largest number M of large clusters ANY cut of
    the dendrogram has, the result is the cut closest to the threshold among those with M large clusters (every strictly better
    candidate replaces the previous one), i.e. exactly M labels; with M = 1 everything collapses into one cluster.  M is found
    here by scanning every cut, independently of either implementation."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from oracle import pyannote_pipeline_ref as O
    rng = np.random.default_rng(11)
    seen = {1: 0, 2: 0, 3: 0}
    for trial in range(1200):
        k = int(rng.integers(1, 4))
        n = int(rng.integers(30, 70))
        cents = rng.standard_normal((k, 16)) * 3.0
        e = (cents[rng.integers(0, k, n)] + rng.standard_normal((n, 16)) * float(rng.choice([0.15, 1.0, 2.0]))).astype(np.float32)
        mcs, target = 12, int(rng.choice([4, 6]))
        big = min(mcs, max(1, round(0.1 * n)))
        en = e.astype(np.float64) / np.linalg.norm(e.astype(np.float64), axis=-1, keepdims=True)
        Z = linkage(en, method="centroid", metric="euclidean")
        Zi = Z.copy(); Zi[:, 2] = np.arange(n - 1)
        M = max(int((np.unique(fcluster(Zi, it, criterion="distance"), return_counts=True)[1] >= big).sum()) for it in range(n - 1))
        if M >= target or M not in seen:
            continue
        seen[M] += 1
        got = P.agglomerative_centroid(e, 0.7045654963945799, mcs, target, target, target)
        want = O.agglomerative_cluster(e, 0.7045654963945799, mcs, target, target, target)
        assert np.array_equal(got, want), trial
        assert len(np.unique(got)) == M, (trial, M, np.unique(got))
    assert seen[2] >= 3 and seen[3] >= 3, seen
