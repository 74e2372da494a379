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
    assert torch.equal(find_sepformer_checkpoint(str(tmp_path))["encoder.conv1d.weight"], sd["encoder.conv1d.weight"].float())
    (ft / "hyperparams.yaml").write_text("# present\n")
    over = find_sepformer_checkpoint(str(tmp_path))
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
    save_file({"decoder.ln.bias": torch.ones(64), "extra.key": torch.zeros(2)}, str(tmp_path / "whisper-ft" / "model.safetensors"))
    _, got = find_whisper_checkpoint("small.en", str(tmp_path))
    assert float(got["decoder.ln.bias"].mean()) == 1.0 and "extra.key" not in got
    assert find_whisper_checkpoint("small.en", str(tmp_path / "absent")) is None
