"""CPU: the file-format loaders of SURVEY 8f-2 (vocabulary files, checkpoint layouts, fine-tune overlays), exercised on
files this test writes itself -- no real vocabulary or checkpoint exists offline."""
import base64
import json
import os

import pytest
import torch

from clearconverse_amd import tokenizer as T
from clearconverse_amd.weights import SepDims, find_sepformer_checkpoint, find_whisper_checkpoint, synthetic_sepformer_state_dict


def _toy_ranks():
    toks = [bytes([b]) for b in range(256)]
    toks += [b" t", b"he", b" th", b" the", b"ll", b"llo", b"hello", b" w", b"or", b" wor", b"ld", b" world", "é".encode()]
    return {t: i for i, t in enumerate(toks)}


def test_tiktoken_bpe_merges_by_rank_and_round_trips(tmp_path):
    ranks = _toy_ranks()
    f = tmp_path / "gpt2.tiktoken"
    f.write_bytes(b"\n".join(base64.b64encode(t) + b" " + str(r).encode() for t, r in ranks.items()))
    tk = T.TiktokenBPE(str(f))
    # " the": bytes ' ', 't', 'h', 'e' -> merges " t" (256) first, then "he" (257) ... lowest rank wins at every round
    assert tk.encode(" the") == [ranks[b" the"]]
    assert tk.encode("hello world") == [ranks[b"hello"], ranks[b" world"]]
    assert tk.encode("hell") == [ranks[b"he"], ranks[b"ll"]]
    text = "hello world, the café 42!"
    assert tk.decode(tk.encode(text)) == text
    os.makedirs(tmp_path / "cache" / "whisper", exist_ok=True)
    (tmp_path / "cache" / "whisper" / "gpt2.tiktoken").write_bytes(f.read_bytes())
    assert isinstance(T.get_tokenizer(str(tmp_path / "cache")), T.TiktokenBPE)
    assert isinstance(T.get_tokenizer(str(tmp_path / "nowhere")), T.IdTokenizer)


def test_gpt2_vocab_json_and_tiktoken_agree(tmp_path):
    ranks = _toy_ranks()
    probe = T.GPT2BPE.__new__(T.GPT2BPE)           # byte <-> unicode table of the GPT-2 files
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    b2u = dict(zip(bs, map(chr, cs)))
    uni = lambda t: "".join(b2u[b] for b in t)
    (tmp_path / "vocab.json").write_text(json.dumps({uni(t): r for t, r in ranks.items()}), encoding="utf-8")
    merges = []
    for t in list(ranks)[256:]:                     # every multi-byte token is the merge of two existing tokens
        for cut in range(1, len(t)):
            if t[:cut] in ranks and t[cut:] in ranks and ranks[t[:cut]] < ranks[t] and ranks[t[cut:]] < ranks[t]:
                merges.append(f"{uni(t[:cut])} {uni(t[cut:])}")
                break
    (tmp_path / "merges.txt").write_text("#version: 0.2\n" + "\n".join(merges) + "\n", encoding="utf-8")
    (tmp_path / "gpt2.tiktoken").write_bytes(b"\n".join(base64.b64encode(t) + b" " + str(r).encode() for t, r in ranks.items()))
    a, b = T.GPT2BPE(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt")), T.TiktokenBPE(str(tmp_path / "gpt2.tiktoken"))
    for text in ["hello world", " the the", "hell or world", "café"]:
        assert a.encode(text) == b.encode(text), text
        assert a.decode(a.encode(text)) == text


def test_sepformer_checkpoint_layout_and_overlay(tmp_path):
    dims = SepDims()
    sd = synthetic_sepformer_state_dict(dims, seed=1)
    base = tmp_path / "resepformer"
    base.mkdir()
    for part in ("encoder", "masknet", "decoder"):
        torch.save({k[len(part) + 1:]: v for k, v in sd.items() if k.startswith(part + ".")}, base / f"{part}.ckpt")
    got = find_sepformer_checkpoint(str(tmp_path))
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k].float()) for k in sd)
    # overlay: applied only when all four files exist; unknown keys ignored (strict=False)
    ft = tmp_path / "resepformer-ft"
    ft.mkdir()
    torch.save({"conv1d.weight": torch.full_like(sd["encoder.conv1d.weight"], 2.0), "not_a_key": torch.zeros(1)}, ft / "encoder.ckpt")
    torch.save({}, ft / "masknet.ckpt")
    torch.save({}, ft / "decoder.ckpt")
    assert torch.equal(find_sepformer_checkpoint(str(tmp_path), apply_ft_overlay=True)["encoder.conv1d.weight"], sd["encoder.conv1d.weight"].float())
    (ft / "hyperparams.yaml").write_text("# present\n")
    # default = the reference's EFFECTIVE behaviour: its load_state_dict({'masknet': .., 'encoder': .., 'decoder': ..}, strict=False)
    # matches no parameter name, so the base weights stay (back/api.py:739-746)
    same = find_sepformer_checkpoint(str(tmp_path))
    assert torch.equal(same["encoder.conv1d.weight"], sd["encoder.conv1d.weight"].float())
    over = find_sepformer_checkpoint(str(tmp_path), apply_ft_overlay=True)      # the intended overlay, behind an explicit switch
    assert float(over["encoder.conv1d.weight"].mean()) == 2.0 and "encoder.not_a_key" not in over
    assert find_sepformer_checkpoint(str(tmp_path / "absent")) is None


def test_whisper_checkpoint_layout_and_overlay(tmp_path):
    from safetensors.torch import save_file
    dims = dict(n_mels=80, n_audio_ctx=1500, n_audio_state=64, n_audio_head=1, n_audio_layer=1, n_vocab=51864, n_text_ctx=448,
                n_text_state=64, n_text_head=1, n_text_layer=1)
    sd = {"encoder.conv1.weight": torch.randn(64, 80, 3), "decoder.ln.bias": torch.randn(64)}
    (tmp_path / "whisper").mkdir()
    torch.save({"dims": dims, "model_state_dict": sd}, tmp_path / "whisper" / "small.en.pt")
    d, got = find_whisper_checkpoint("small.en", str(tmp_path))
    assert d.n_audio_state == 64 and torch.equal(got["decoder.ln.bias"], sd["decoder.ln.bias"])
    (tmp_path / "whisper-ft").mkdir()
    save_file({"decoder.ln.bias": torch.ones(64), "extra.key": torch.zeros(2), "encoder.conv1.weight": torch.zeros(8, 8)},
              str(tmp_path / "whisper-ft" / "model.safetensors"))
    _, got = find_whisper_checkpoint("small.en", str(tmp_path))
    assert float(got["decoder.ln.bias"].mean()) == 1.0 and "extra.key" not in got
    assert torch.equal(got["encoder.conv1.weight"], sd["encoder.conv1.weight"])      # a wrong-shaped overlay tensor is not copied
    assert find_whisper_checkpoint("small.en", str(tmp_path / "absent")) is None


class _Spec:
    """Stands in for pyannote.audio.core.task.Specifications (a pickled custom object inside the published checkpoints)."""


def _hub(tmp_path, sub, repo, rev="abc123"):
    org, name = repo.split("/")
    d = tmp_path / sub / f"models--{org}--{name}" / "snapshots" / rev
    d.mkdir(parents=True)
    return d


def test_pyannote_checkpoints_and_pipeline_configs_from_the_hub_cache_layout(tmp_path, monkeypatch):
    """The three pyannote-side constructors of the reference (back/api.py:776-792) leave Lightning checkpoints
    (`pytorch_model.bin` = {"state_dict": ...}) and pipeline `config.yaml` files in hub-cache directories.  Written here in
    that layout with seeded tensors: the loaders must find them, return exactly the tensors, read the hyper-parameters, and
    `build_state_dicts` must use them instead of synthetic weights."""
    from clearconverse_amd import weights as W
    from clearconverse_amd.models import build_state_dicts
    monkeypatch.setenv("MODEL_CACHE_DIR", str(tmp_path))
    for env in ("PYANNOTE_CACHE", "HF_HOME", "HUGGINGFACE_HUB_CACHE", "HF_HUB_CACHE"):
        monkeypatch.delenv(env, raising=False)
    monkeypatch.setenv("HOME", str(tmp_path / "home"))
    assert W.find_pyannote_checkpoint("xvector") is None and W.find_pipeline_config("vad")["source"] == "defaults"

    xv = W.synthetic_xvector_state_dict(seed=21)
    extra = dict(xv); extra["sincnet.conv1d.0.filterbank.window_"] = torch.zeros(125)          # upstream buffers the kernels do not take
    torch.save({"state_dict": extra, "pytorch-lightning_version": "1.6.5", "hyper_parameters": {"sample_rate": 16000}},
               _hub(tmp_path, "embedding", "pyannote/embedding") / "pytorch_model.bin")
    seg3 = W.synthetic_pyannet_state_dict(3, seed=22)
    torch.save({"state_dict": seg3}, _hub(tmp_path, "vad", "pyannote/segmentation") / "pytorch_model.bin")
    seg7 = W.synthetic_pyannet_state_dict(7, seed=23)
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in seg7.items()}, str(_hub(tmp_path, "speaker-diarization", "pyannote/segmentation-3.0") / "model.safetensors"))
    rn = W.synthetic_resnet34_state_dict(seed=24)
    torch.save(rn, _hub(tmp_path, "speaker-diarization", "pyannote/wespeaker-voxceleb-resnet34-LM") / "pytorch_model.bin")    # plain state_dict
    (_hub(tmp_path, "vad", "pyannote/voice-activity-detection") / "config.yaml").write_text(
        "pipeline:\n  name: pyannote.audio.pipelines.VoiceActivityDetection\n  params:\n    segmentation: pyannote/segmentation\n"
        "params:\n  onset: 0.5\n  offset: 0.25\n  min_duration_on: 0.1\n  min_duration_off: 0.2\n")
    (_hub(tmp_path, "speaker-diarization", "pyannote/speaker-diarization-3.1") / "config.yaml").write_text(
        "version: 3.1.0\npipeline:\n  name: pyannote.audio.pipelines.SpeakerDiarization\n  params:\n    clustering: AgglomerativeClustering\n"
        "    embedding: pyannote/wespeaker-voxceleb-resnet34-LM\n    segmentation: pyannote/segmentation-3.0\n"
        "params:\n  clustering:\n    method: centroid\n    min_cluster_size: 9\n    threshold: 0.61\n  segmentation:\n    min_duration_off: 0.05\n")

    got = W.find_pyannote_checkpoint("xvector")
    assert set(got) == set(xv) and all(torch.equal(got[k], xv[k]) for k in xv)
    for kind, want in (("pyannet_vad", seg3), ("pyannet_diar", seg7), ("resnet34", rn)):
        got = W.find_pyannote_checkpoint(kind)
        assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want), kind
    vad = W.find_pipeline_config("vad")
    assert (vad["onset"], vad["offset"], vad["min_duration_on"], vad["min_duration_off"]) == (0.5, 0.25, 0.1, 0.2)
    assert vad["models"]["segmentation"] == "pyannote/segmentation" and vad["source"].endswith("config.yaml")
    di = W.find_pipeline_config("diarization")
    assert (di["threshold"], di["min_cluster_size"], di["min_duration_off"], di["method"]) == (0.61, 9, 0.05, "centroid")
    sds = build_state_dicts(None, whisper_dims=W.WhisperDims.mini(1, 64), sep_dims=SepDims(n_layers=1))
    assert sds["weights_sources"]["xvector"] == "checkpoint" and sds["weights_sources"]["resnet34"] == "checkpoint"
    assert torch.equal(sds["pyannet_vad"]["classifier.weight"], seg3["classifier.weight"]) and sds["vad_params"]["onset"] == 0.5
    assert sds["diarization_params"]["threshold"] == 0.61


def test_pyannote_loader_fails_loudly_on_wrong_architecture_and_reports_refused_files(tmp_path, monkeypatch):
    from clearconverse_amd import weights as W
    monkeypatch.setenv("MODEL_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("HOME", str(tmp_path / "home"))
    bad = W.synthetic_pyannet_state_dict(7, seed=1)
    del bad["lstm.weight_hh_l3_reverse"]
    torch.save({"state_dict": bad}, _hub(tmp_path, "speaker-diarization", "pyannote/segmentation-3.0") / "pytorch_model.bin")
    with pytest.raises(ValueError, match="lacks 'lstm.weight_hh_l3_reverse'"):
        W.find_pyannote_checkpoint("pyannet_diar")
    # a checkpoint carrying a pickled custom object (as pyannote's published files do): the weights-only loader refuses it, nothing
    # from the file is executed, the refusal is reported and the caller falls back
    d = _hub(tmp_path, "embedding", "pyannote/embedding")
    torch.save({"state_dict": {}, "pyannote.audio": {"specifications": _Spec()}}, d / "pytorch_model.bin")
    said = []
    assert W.find_pyannote_checkpoint("xvector", log=said.append) is None
    assert said and "refused" in said[0]
    with pytest.raises(ValueError, match="not implemented"):
        (_hub(tmp_path, "speaker-diarization", "pyannote/speaker-diarization-3.1") / "config.yaml").write_text(
            "params:\n  clustering:\n    method: average\n    min_cluster_size: 12\n    threshold: 0.7\n")
        W.find_pipeline_config("diarization")


def test_conform_accepts_size_one_dimensions_only(tmp_path, monkeypatch):
    """a checkpoint tensor whose shape differs from the architecture's may be reshaped only when nothing but size-1
    dimensions differ (element order unchanged); a transposed / permuted weight of the same numel is an error, not a silent reshape."""
    from clearconverse_amd import weights as W
    schema = {"a": torch.zeros(4, 6), "b": torch.zeros(5), "c": torch.zeros(3, 1, 7)}
    ok = W._conform("x", {"a": torch.arange(24.).reshape(4, 6, 1), "b": torch.arange(5.).reshape(1, 5), "c": torch.arange(21.).reshape(3, 7)}, schema, "f")
    assert ok["a"].shape == (4, 6) and ok["b"].shape == (5,) and ok["c"].shape == (3, 1, 7) and torch.equal(ok["a"].flatten(), torch.arange(24.))
    with pytest.raises(ValueError, match="has shape"):
        W._conform("x", {"a": torch.zeros(6, 4), "b": torch.zeros(5), "c": torch.zeros(3, 1, 7)}, schema, "f")      # transposed
    with pytest.raises(ValueError, match="lacks"):
        W._conform("x", {"a": torch.zeros(4, 6)}, schema, "f")
