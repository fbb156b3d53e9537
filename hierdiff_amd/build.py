"""Build libhierdiff_hip.so (gfx950) in-tree with hipcc.

    python -m hierdiff_amd.build [--force] [--save-temps]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libhierdiff_hip.so")
SOURCES = [os.path.join(CSRC, "hierdiff_hip.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")] + [
    os.path.join(os.path.dirname(PKG), "include", "hierdiff_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-o", LIB + ".tmp"] + SOURCES
    if save_temps:
        tmpdir = os.path.join(PKG, "build")
        os.makedirs(tmpdir, exist_ok=True)
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(LIB)
