"""-m gpu: SURVEY 8f-2 on the device -- checkpoint FILES in the upstream layouts -> `load_models(Config())` -> HIP outputs.

The reference's loaders (/root/reference/back/api.py:665-692 Whisper + `whisper-ft/model.safetensors` overlay, 713-746 SpeechBrain
savedir + `resepformer-ft`, 776-792 the three pyannote constructors) leave these files under MODEL_CACHE_DIR.  No real file exists
offline, so the test writes seeded weights in those layouts (openai-whisper `.pt` with a `dims` header, `encoder/masknet/decoder.ckpt`,
Lightning `pytorch_model.bin` / safetensors in hub-cache directories, the pipelines' `config.yaml`), points MODEL_CACHE_DIR at them and
requires every network built by `load_models(Config())` to give BIT-IDENTICAL outputs to the same networks built from the in-memory
state dicts (`state_dicts=`): a key dropped, transposed, overlaid wrongly or read at another precision on the way from disk to
HBM shows up as a different output.  (Text <-> ids parity stays unpinned: the GPT-2 vocabulary is not on disk.)"""
import numpy as np
import pytest
import torch

from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.weights import SepDims, WhisperDims

pytestmark = pytest.mark.gpu


def _hub(root, sub, repo, rev="0123abcd"):
    org, name = repo.split("/")
    d = root / sub / f"models--{org}--{name}" / "snapshots" / rev
    d.mkdir(parents=True)
    return d


def test_checkpoint_files_reach_the_hip_models_unchanged(ccx_ctx, tmp_path, monkeypatch):
    from safetensors.torch import save_file
    from clearconverse_amd.models import build_state_dicts, load_models
    from clearconverse_amd.processor import Config
    monkeypatch.setenv("HOME", str(tmp_path / "home"))
    for env in ("PYANNOTE_CACHE", "HF_HOME", "HUGGINGFACE_HUB_CACHE", "HF_HUB_CACHE", "CCX_APPLY_RESEPFORMER_FT"):
        monkeypatch.delenv(env, raising=False)
    monkeypatch.setenv("MODEL_CACHE_DIR", str(tmp_path / "empty"))
    wd = WhisperDims.mini(2, 128)
    want = build_state_dicts(None, whisper_dims=wd, sep_dims=SepDims(), seed=31)        # in-memory truth (synthetic: the cache is empty)
    assert all(v.startswith("synthetic") for v in want["weights_sources"].values())

    root = tmp_path / "cache"
    # Whisper: <cache>/whisper/small.en.pt = {"dims", "model_state_dict"} (openai-whisper), fp16 on disk like the published files
    (root / "whisper").mkdir(parents=True)
    base = {k: v.clone() for k, v in want["whisper"].items()}
    torch.save({"dims": dict(wd.__dict__), "model_state_dict": {k: v.half() for k, v in base.items()}}, root / "whisper" / "small.en.pt")
    # ... + the fine-tune overlay (reference 671-692): two tensors replaced, an unknown key and a wrong-shaped one ignored
    (root / "whisper-ft").mkdir()
    g = torch.Generator().manual_seed(5)
    over = {"decoder.ln.bias": torch.randn(wd.n_text_state, generator=g), "encoder.blocks.1.mlp.0.weight": 0.05 * torch.randn(4 * wd.n_audio_state, wd.n_audio_state, generator=g)}
    save_file({**over, "not.a.key": torch.zeros(3), "encoder.conv1.weight": torch.zeros(4, 4)}, str(root / "whisper-ft" / "model.safetensors"))
    want["whisper"] = {k: v.half().float() for k, v in base.items()}
    want["whisper"].update({k: v.clone() for k, v in over.items()})
    # RE-SepFormer: SpeechBrain savedir (reference 713-727)
    (root / "resepformer").mkdir()
    for part in ("encoder", "masknet", "decoder"):
        torch.save({k[len(part) + 1:]: v for k, v in want["sepformer"].items() if k.startswith(part + ".")}, root / "resepformer" / f"{part}.ckpt")
    # pyannote side (reference 776-792): hub-cache layout, Lightning checkpoints / plain / safetensors
    torch.save({"state_dict": want["xvector"], "pytorch-lightning_version": "1.6.5"}, _hub(root, "embedding", "pyannote/embedding") / "pytorch_model.bin")
    torch.save({"state_dict": want["pyannet_vad"]}, _hub(root, "vad", "pyannote/segmentation") / "pytorch_model.bin")
    save_file({k: v.contiguous() for k, v in want["pyannet_diar"].items()}, str(_hub(root, "speaker-diarization", "pyannote/segmentation-3.0") / "model.safetensors"))
    torch.save(want["resnet34"], _hub(root, "speaker-diarization", "pyannote/wespeaker-voxceleb-resnet34-LM") / "pytorch_model.bin")
    (_hub(root, "vad", "pyannote/voice-activity-detection") / "config.yaml").write_text(
        "pipeline:\n  name: pyannote.audio.pipelines.VoiceActivityDetection\n  params:\n    segmentation: pyannote/segmentation\n"
        "params:\n  onset: 0.6\n  offset: 0.3\n  min_duration_on: 0.11\n  min_duration_off: 0.09\n")
    (_hub(root, "speaker-diarization", "pyannote/speaker-diarization-3.1") / "config.yaml").write_text(
        "version: 3.1.0\npipeline:\n  name: pyannote.audio.pipelines.SpeakerDiarization\n  params:\n    clustering: AgglomerativeClustering\n"
        "    embedding: pyannote/wespeaker-voxceleb-resnet34-LM\n    segmentation: pyannote/segmentation-3.0\n"
        "params:\n  clustering:\n    method: centroid\n    min_cluster_size: 10\n    threshold: 0.65\n  segmentation:\n    min_duration_off: 0.02\n")

    monkeypatch.setenv("MODEL_CACHE_DIR", str(root))
    disk = load_models(Config(), 0, whisper_batch=4, ctx=ccx_ctx, sep_tokens=60_000, max_crops=32, max_audio_seconds=30.0)
    assert disk["weights_sources"] == {k: "checkpoint" for k in ("whisper", "sepformer", "xvector", "pyannet_diar", "pyannet_vad", "resnet34")}
    vp, dp = disk["vad_pipeline"], disk["diarization"]
    assert (vp.onset, vp.offset, vp.min_on, vp.min_off) == (0.6, 0.3, 0.11, 0.09)
    assert (dp.threshold, dp.min_cluster_size, dp.min_off) == (0.65, 10, 0.02)
    mem = load_models(None, 0, whisper_batch=4, ctx=ccx_ctx, state_dicts=want, sep_tokens=60_000, max_crops=32, max_audio_seconds=30.0)
    try:
        clip = synthetic_clip(3, 30.0)
        dev = torch.from_numpy(clip[None]).cuda()
        outs = []
        for m in (disk, mem):
            w = m["whisper_model"]
            assert w.dims == wd
            mel = w.log_mel(dev, [len(clip)], return_mel=True).clone()
            xa = w.encode(1, return_xa=True).clone()
            rec = w.decode_greedy([[w.rules.sot]], sample_len=6)[0]
            sep = m["separator"].separate_batch(dev[:, : 16000 * 3].contiguous(), [16000 * 3]).clone()
            emb = m["embedding_model"].embed_batch([dev[0, : 16000 * 2], dev[0, 16000 * 5: 16000 * 9]]).clone()
            seg_v = m["segmentation_vad"].segment_numpy([dev[0, :80000]])[0]
            seg_d = m["segmentation_diar"].segment_numpy([dev[0, :160000]])[0]
            res = m["diarization_embedder"].embed_chunks(dev[:, :160000].contiguous()).clone()
            outs.append(dict(mel=mel, xa=xa, tokens=rec["tokens"], lp=rec["sum_logprob"], sep=sep, emb=emb, seg_v=seg_v, seg_d=seg_d, res=res))
        a, b = outs
        for k in ("mel", "xa", "sep", "emb", "res"):
            assert torch.equal(a[k], b[k]), k
            assert bool(torch.isfinite(a[k]).all()) and float(a[k].abs().max()) > 0, k
        assert np.array_equal(a["seg_v"], b["seg_v"]) and np.array_equal(a["seg_d"], b["seg_d"])
        assert a["tokens"] == b["tokens"] and a["lp"] == b["lp"] and len(a["tokens"]) > 0
        # the overlay really is in the device copy: the same network WITHOUT it gives another encoder output
        want["whisper"] = {k: v.half().float() for k, v in base.items()}
        plain = load_models(None, 0, whisper_batch=4, ctx=ccx_ctx, state_dicts=want, sep_tokens=60_000, max_crops=32, max_audio_seconds=30.0)
        plain["whisper_model"].log_mel(dev, [len(clip)])
        assert not torch.equal(plain["whisper_model"].encode(1, return_xa=True), a["xa"])
        for k in ("whisper_model", "separator", "embedding_model", "diarization_embedder", "segmentation_vad", "segmentation_diar", "denoiser"):
            plain[k].close()
    finally:
        for m in (disk, mem):
            for k in ("whisper_model", "separator", "embedding_model", "diarization_embedder", "segmentation_vad", "segmentation_diar", "denoiser"):
                m[k].close()
