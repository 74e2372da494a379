"""-m gpu: A13 / A14 -- `VoiceActivityDetection` and `SpeakerDiarization` (clearconverse_amd/pipelines.py on the libccx networks)
against the oracle pipelines (oracle networks -> oracle/pyannote_pipeline_ref.py post-net); reference calls
/root/reference/back/api.py:1311-1312 (VAD), 1052-1064 and 1120-1137 (diarization, min_speakers=1 / max_speakers=2).

Seeded random segmentation weights give one constant class, so the segmentation networks carry SCRIPTED weights
(tests/scripted_nets.py: fitted on the CPU oracle so that the outputs follow the synthetic clip's schedule -- silences, two
speakers, an overlap, realistic flicker at syllable troughs).  Parity unpinned: the oracle is a recalled restatement.

Three levels:
 (1) the post-net in isolation on REAL network outputs: the oracle post-net fed with the GPU's own frame scores and embeddings
     must return exactly the product's Annotation (same labels, same boundaries to the last bit);
 (2) network decisions: the hard multi-label activity decoded from GPU and oracle scores agrees on >= 99.9 % of the frames (measured 99.97 %),
     embeddings of the oracle's masks agree to rel-L2 8e-3 (measured 3.1e-3);
 (3) end to end (HIP nets + product post-net vs oracle nets + oracle post-net): VAD region count equal and boundaries within
     one frame (270 samples); diarization timelines agree on >= 99.85 % of the (time, speaker) cells (measured 99.94 %) up to a label permutation --
     a frame whose top-2 margin is below the bf16 error may flip a one-frame segment, and the clustering of seeded ResNet
     embeddings (cosine ~0.99 between any two) is not a stable function of its input bits.
Bounds are <= 2.5x the worst deviation measured on MI355X (profiles/r03_measured_deviations.json); tests.conftest.within records
every measured value again on each run."""
import numpy as np
import pytest
import torch

from tests.conftest import within

from clearconverse_amd.audio import synthetic_clip

pytestmark = pytest.mark.gpu
FRAME = 270 / 16000


@pytest.fixture(scope="module")
def nets(ccx_ctx):
    from clearconverse_amd.speaker import ResNetEmbedder, SegmentationNet
    from clearconverse_amd.weights import synthetic_resnet34_state_dict
    from tests.scripted_nets import scripted_pyannet_state_dict
    sd_diar, rep_d = scripted_pyannet_state_dict(1, 7, True)
    sd_vad, rep_v = scripted_pyannet_state_dict(1, 3, False, window_s=5.0, seed=4)
    assert rep_d["accuracy"] > 0.9 and rep_v["accuracy"] > 0.95, (rep_d, rep_v)
    rsd = synthetic_resnet34_state_dict(seed=5)
    seg_d = SegmentationNet(sd_diar, n_classes=7, powerset=True, max_crops=64, max_samples=16000 * 400, ctx=ccx_ctx)
    seg_v = SegmentationNet(sd_vad, n_classes=3, powerset=False, max_crops=64, max_samples=16000 * 400, ctx=ccx_ctx)
    emb = ResNetEmbedder(rsd, max_chunks=32, max_samples=160000, max_masks=512, ctx=ccx_ctx)
    yield dict(sd_diar=sd_diar, sd_vad=sd_vad, rsd=rsd, seg_d=seg_d, seg_v=seg_v, emb=emb)
    for m in (seg_d, seg_v, emb):
        m.close()


def _clips():
    full = synthetic_clip(1, 30.0)
    return {"30 s": full, "23.7 s (zero-padded last windows)": full[: int(23.7 * 16000)].copy()}


def _tracks(ann):
    return [(s.start, s.end, l) for s, _, l in ann.itertracks(yield_label=True)]


def test_postnet_equals_oracle_postnet_on_gpu_network_outputs(nets):
    from clearconverse_amd import pipelines as P
    from oracle import pyannote_pipeline_ref as O
    vad = P.VoiceActivityDetection(nets["seg_v"])
    dia = P.SpeakerDiarization(nets["seg_d"], nets["emb"])
    for name, clip in _clips().items():
        item = {"waveform": torch.from_numpy(clip), "sample_rate": 16000}
        got = [(s, e) for s, e, _ in _tracks(vad(item))]
        want = O.voice_activity_detection(clip, lambda ch: nets["seg_v"].segment_numpy([torch.from_numpy(c) for c in ch]), False)
        assert got == want and len(got) >= 2, name

        def emb_fn(chunk, mask):
            return nets["emb"].embed_chunks(torch.from_numpy(chunk)[None], torch.from_numpy(mask)[None], [0])[0].cpu().numpy()
        for kw in (dict(min_speakers=1, max_speakers=2), dict(min_speakers=2, max_speakers=2)):
            got = _tracks(dia(item, **kw))
            want = sorted(O.speaker_diarization(clip, lambda ch: nets["seg_d"].segment_numpy([torch.from_numpy(c) for c in ch]), emb_fn, **kw),
                          key=lambda t: (t[0], t[1]))
            assert got == want, (name, kw)
            assert len(got) >= 4 and (kw["min_speakers"] == 1 or len({l for *_, l in got}) == 2), (name, kw)


def test_network_decisions_match_the_oracle_networks(nets):
    from oracle import pyannote_pipeline_ref as O
    from oracle import wespeaker_ref as W
    from tests.pinned_oracle import oracle_seg_fn
    clip = synthetic_clip(1, 30.0)
    starts, chunks = O.cut_chunks(clip, 160000, 16000)
    gpu = nets["seg_d"].segment_numpy([torch.from_numpy(c) for c in chunks])
    orc = oracle_seg_fn(nets["sd_diar"], True)(chunks)
    agree = np.mean([np.mean(O.powerset_to_multilabel(g) == O.powerset_to_multilabel(o)) for g, o in zip(gpu, orc)])
    worst = max(float(np.abs(g - o).max()) for g, o in zip(gpu, orc))
    # the fitted classifier has a gain of ~6 per unit of the (unit-variance) features, so log-probabilities span +-12 and their absolute
    # error is ~10x that of the seeded-weight test (test_speaker_gpu.py: 5e-2); what matters here are the DECISIONS: identical wherever
    # the oracle's top-2 margin exceeds 1.0, and >= 99.5 % of all frames
    print(f"scripted segmentation: hard multi-label agreement {agree:.4f}, max |log-prob error| {worst:.3e}")
    within("pyannet scripted weights: 1 - hard multi-label agreement with the oracle", 1.0 - agree, 1e-3)
    within("pyannet scripted weights: log-prob max abs error (logit span +-12)", worst, 0.45)
    for g, o in zip(gpu, orc):
        top2 = np.sort(o, axis=-1)[:, -2:]
        decided = (top2[:, 1] - top2[:, 0]) > 1.0
        assert np.array_equal(g.argmax(-1)[decided], o.argmax(-1)[decided])
    _, ch5 = O.cut_chunks(clip, 80000, 8000)
    gv = nets["seg_v"].segment_numpy([torch.from_numpy(c) for c in ch5])
    ov = oracle_seg_fn(nets["sd_vad"], False)(ch5)
    within("pyannet scripted weights (VAD, sigmoid): frame score max abs error", max(float(np.abs(g - o).max()) for g, o in zip(gv, ov)), 3e-2)
    # embeddings of the ORACLE's masks (the pooling masks the pipeline would hand over) for four windows
    rel = 0.0
    for c in (0, 7, 12, 20):
        masks = np.stack(O.embedding_masks(O.powerset_to_multilabel(orc[c]), 2))
        masks = masks[masks.sum(1) > 0]
        got = nets["emb"].embed_chunks(torch.from_numpy(chunks[c])[None], torch.from_numpy(masks), [0] * len(masks)).cpu().numpy()
        ref = W.resnet_embed(nets["rsd"], chunks[c][None], masks, np.zeros(len(masks), dtype=np.int64))
        rel = max(rel, float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
    print(f"ResNet-34 embeddings of the oracle's pooling masks: worst rel-L2 {rel:.3e}")
    within("resnet34: embedding rel-L2 (pipeline pooling masks, 10 s chunks)", rel, 8e-3)


def test_pipelines_end_to_end_against_the_oracle_pipelines(nets):
    from clearconverse_amd import pipelines as P
    from tests import pinned_oracle as PO
    sds = dict(pyannet_vad=nets["sd_vad"], pyannet_diar=nets["sd_diar"], resnet34=nets["rsd"])
    vad = P.VoiceActivityDetection(nets["seg_v"])
    dia = P.SpeakerDiarization(nets["seg_d"], nets["emb"])
    for name, clip in _clips().items():
        item = {"waveform": torch.from_numpy(clip), "sample_rate": 16000}
        want = PO.run_pipelines(clip, sds, min_speakers=1, max_speakers=2)
        got_v = [(s, e) for s, e, _ in _tracks(vad(item))]
        assert len(got_v) == len(want["vad"]) >= 2, (name, got_v, want["vad"])
        dev = max(max(abs(a[0] - b[0]), abs(a[1] - b[1])) for a, b in zip(got_v, want["vad"]))
        within("pipelines end to end: VAD boundary deviation (s)", dev, FRAME + 1e-9, name)
        got_d = _tracks(dia(item, min_speakers=1, max_speakers=2))
        agree = PO.timeline_agreement(got_d, want["diarization"], len(clip) / 16000 + 10.0)
        n_lab = (len({l for *_, l in got_d}), len({l for *_, l in want["diarization"]}))
        print(f"{name}: VAD {len(got_v)} regions, worst boundary deviation {dev * 1e3:.1f} ms; diarization {len(got_d)} vs "
              f"{len(want['diarization'])} turns, labels {n_lab}, agreement {agree:.4f}")
        within("pipelines end to end: diarization timeline disagreement with the oracle pipeline", 1.0 - agree, 1.5e-3, name)
        assert n_lab[0] == n_lab[1], (name, n_lab)
