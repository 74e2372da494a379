import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracles are chains of small torch ops (a 589-step LSTM loop, 3 x 3 convolutions on short chunks): on the GPU box's
    # 256 host cores torch's default thread count makes them several times SLOWER than on 8-16 threads (the two pipeline parity
    # tests took 160 s and 90 s).  Importing torch does not touch the GPU.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def ccx_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from clearconverse_amd import _lib
    return _lib.Context(0)


# ---- measured deviations ------------------------------------------------------------------------------------------------
# Every parity assertion against an oracle goes through `within(name, value, bound)`: it asserts value < bound and keeps the
# worst value seen per name.  At the end of a GPU session the table is printed and written to gpurun_out/measured_deviations.json
# (copied to profiles/ per round); DESIGN.md section 3 lists each bound next to the measured worst (bounds are <= 2x measured
# unless justified there).
_DEVIATIONS = {}


def within(name: str, value, bound: float, ctx=None) -> None:
    value = float(value)
    rec = _DEVIATIONS.setdefault(name, {"worst": 0.0, "bound": float(bound), "checks": 0})
    rec["worst"] = max(rec["worst"], value)
    rec["bound"] = max(rec["bound"], float(bound))
    rec["checks"] += 1
    assert value < bound, (name, value, bound, ctx)


def pytest_sessionfinish(session, exitstatus):
    if not _DEVIATIONS:
        return
    import json
    rows = {k: dict(v, ratio=round(v["worst"] / v["bound"], 3)) for k, v in sorted(_DEVIATIONS.items())}
    print("\nmeasured worst deviations (name: worst / bound):")
    for k, v in rows.items():
        print(f"  {k:70s} {v['worst']:.3e} / {v['bound']:.1e}  ({v['checks']} checks)")
    out = ROOT / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "measured_deviations.json").write_text(json.dumps(rows, indent=1))
    except OSError:
        pass
