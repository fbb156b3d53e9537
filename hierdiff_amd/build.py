"""Build libhierdiff_hip.so (gfx950) in-tree with hipcc.

    python -m hierdiff_amd.build [--force] [--save-temps] [--debug-kernels]

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


RESOURCES = os.path.join(LIBDIR, "kernel_resources.json")


def _parse_resource_remarks(text: str) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage -> {mangled kernel: {VGPRs, AGPRs, ScratchSize, VGPRs Spill, ...}}."""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z /\[\]]*?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            key = m.group(1).strip()
            try:
                cur[key] = int(m.group(2))
            except ValueError:
                cur[key] = m.group(2)
    return out


def _production(name: str) -> bool:
    """Everything except the ablated / traced edge-kernel instantiations of a --debug-kernels build."""
    import re
    m = re.match(r"_Z6k_edgeILi\d+ELb[01]ELi[0123]ELi(\d+)EE", name)
    if m:
        return m.group(1) == "0"
    return True


def audit(resources: dict) -> list:
    """Kernels hide loads from the compiler (inline-asm loads released by hand-counted s_waitcnt): a register spill
    next to one of them would save a destination before its data has landed.  No production kernel may spill."""
    bad = []
    for name, r in resources.items():
        if not name.startswith("_Z") or not _production(name):
            continue
        if r.get("ScratchSize", 0) or r.get("VGPRs Spill", 0):          # SGPR spills go to VGPR lanes, not to memory
            bad.append(f"{name}: scratch {r.get('ScratchSize')} B/lane, VGPR spills {r.get('VGPRs Spill')}")
    return bad


LIB_DEBUG = os.path.join(LIBDIR, "libhierdiff_hip_dbg.so")     # --debug-kernels build; select with HIERDIFF_LIB=<path>


def build(force: bool = False, save_temps: bool = False, verbose: bool = True, debug_kernels: bool = False) -> str:
    if debug_kernels:
        return _build_debug(verbose)
    if not force and not needs_build() and os.path.exists(RESOURCES):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
           "-Rpass-analysis=kernel-resource-usage", "-o", LIB + ".tmp"] + SOURCES
    if save_temps:
        tmpdir = os.path.join(PKG, "build")
        os.makedirs(tmpdir, exist_ok=True)
        cmd += ["-save-temps=obj"]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    other = [ln for ln in proc.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln
             and not ln.startswith("In file included from")]
    if other and (verbose or proc.returncode):
        print("\n".join(other), file=sys.stderr, flush=True)
    if proc.returncode:
        raise subprocess.CalledProcessError(proc.returncode, cmd)
    resources = _parse_resource_remarks(proc.stderr)
    import json
    with open(RESOURCES, "w") as fh:
        json.dump(resources, fh, indent=1, sort_keys=True)
    bad = audit(resources)
    if bad:
        raise RuntimeError("register spills in production kernels (unsafe next to inline-asm loads):\n  " + "\n  ".join(bad))
    os.replace(LIB + ".tmp", LIB)
    return LIB


def _build_debug(verbose: bool = True) -> str:
    """Measurement build next to the product library: HD_ABLATE variants of the edge kernel + hd_debug_edge_trace."""
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [_hipcc(), "-DHD_DEBUG_KERNELS", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC",
           "-shared", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-o", LIB_DEBUG] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB_DEBUG


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv,
                debug_kernels="--debug-kernels" in sys.argv))
