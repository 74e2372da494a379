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
    out = P.aggregate([a, b], [0, 5 * 270], 15 * 270, 10 * 270)
    assert np.allclose(out[:5, 0], 1) and np.allclose(out[5:10, 0], 2) and np.allclose(out[10:15, 0], 3)


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
        merged = []
        for s, e in regions:
            if merged and s - merged[-1][1] < min_off:
                merged[-1] = (merged[-1][0], e)
            else:
                merged.append((s, e))
        return [(s, e) for s, e in merged if e - s >= min_on and e > s]

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
    a = P.aggregate([np.ascontiguousarray(cm[:, w].T) for w in range(6)], starts, 13500 + 40 * 270, 40 * 270)
    b = P.aggregate_cm(cm, starts, 13500 + 40 * 270)
    assert np.array_equal(a, b)


def test_multilabel_cm_equals_powerset_to_multilabel():
    rng = np.random.default_rng(6)
    lp = rng.standard_normal((4, 50, 7)).astype(np.float32)
    want = np.moveaxis(P.powerset_to_multilabel(lp), -1, 0)
    assert np.array_equal(P.multilabel_cm(lp, True), want)
    sc = rng.random((4, 50, 3)).astype(np.float32)
    assert np.array_equal(P.multilabel_cm(sc, False), np.moveaxis((sc > 0.5).astype(np.float32), -1, 0))
