"""Build libccx.so (HIP C++ for gfx950) in-tree with hipcc.

`python -m clearconverse_amd.build` compiles every csrc/*.hip to an object (cached on mtime of the
source and of every header) and links clearconverse_amd/libccx.so.  hipcc cross-compiles without
a GPU, so this runs in the CPU-only build container and the .so travels to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "build"
LIB = PKG / "libccx.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wno-pass-failed", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newest_header() -> float:
    hs = list(CSRC.glob("*.h")) + list((PKG.parent / "include").glob("*.h"))
    return max(h.stat().st_mtime for h in hs)


def build(verbose: bool = True, force: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError(f"no HIP sources under {CSRC}")
    hdr = _newest_header()
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        o = OBJ / (s.stem + ".o")
        if force or not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, hdr):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc, *FLAGS, "-c", str(s), "-o", str(o)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s.name}:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[ccx build] compiled {s.name}", flush=True)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [OBJ / (s.stem + ".o") for s in srcs]
    if jobs or not LIB.exists() or LIB.stat().st_mtime < max(o.stat().st_mtime for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[ccx build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
