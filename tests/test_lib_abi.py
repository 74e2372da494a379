"""CPU: the C-ABI shared library builds, loads and exports every symbol include/ccx.h declares
(no compute calls without a GPU), and the product path fails loudly when no GPU is present."""
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "ccx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ccx_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from clearconverse_amd.build import build
    from clearconverse_amd import _lib
    lib_path = build(verbose=False)
    assert lib_path.exists()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"libccx.so does not export {n}"
        assert n in _lib.PROTOTYPES, f"ctypes binding lacks a prototype for {n}"
    assert set(_lib.PROTOTYPES) == set(names), set(_lib.PROTOTYPES) ^ set(names)
    assert lib.ccx_version().startswith(b"ccx")


def test_no_gpu_fails_loudly():
    # checked inside the test, not in a decorator: a decorator would initialise HIP while pytest COLLECTS, and
    # tests/test_00_service_fork_gpu.py needs a parent process that has not touched the GPU
    if torch.cuda.is_available():
        pytest.skip("checks the no-GPU failure mode")
    from clearconverse_amd import _lib
    with pytest.raises(_lib.CcxError) as e:
        _lib.Context(0)
    assert "HIP device" in str(e.value) or "failed" in str(e.value)
    from clearconverse_amd.weights import WhisperDims
    from clearconverse_amd.whisper import WhisperModel
    with pytest.raises(_lib.CcxError):
        WhisperModel(WhisperDims.mini(), {})


def test_product_path_never_imports_oracle():
    for f in (ROOT / "clearconverse_amd").rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f
        # citations in docstrings are fine; a string literal path would mean the product reads the reference
        assert not re.search(r"[\"']/root/reference", src), f


def test_tokenizer_codec_roundtrip_and_constants():
    from clearconverse_amd import tokenizer as T
    tk = T.IdTokenizer()
    ids = [50363, 464, 1917, 318, 50413]
    assert tk.encode(tk.decode(ids)) == [464, 1917, 318]          # timestamps are dropped from text
    p1 = tk.encode(" This is a conversation between two people.")
    assert p1 == tk.encode(" This is a conversation between two people.") and len(p1) == 7
    assert all(0 <= t < T.EOT for t in p1)
    assert T.TIMESTAMP_BEGIN + 1501 == 51864
    assert set([T.SOT, T.SOT_PREV, T.NO_SPEECH, T.TRANSCRIBE, T.TRANSLATE, T.SOT_LM]) <= set(T.SUPPRESS_TOKENS)
    assert T.EOT not in T.SUPPRESS_TOKENS and T.BLANK not in T.SUPPRESS_TOKENS
