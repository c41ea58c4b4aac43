"""Build libvsde_hip.so (the gfx950 C-ABI library) in-tree with hipcc.

``python -m viforsdes_amd.build`` or ``viforsdes_amd.build.build_library()``.  hipcc
cross-compiles without a GPU; the built .so travels with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "libvsde_hip.so")
SOURCES = ["vsde_gemm.hip", "vsde_head.hip", "vsde_elbo.hip", "vsde_encoder.hip", "vsde_wgrad.hip", "vsde_attn.hip", "vsde_sde.hip"]
HEADERS = ["vsde_common.h", os.path.join("..", "..", "include", "vsde_hip.h")]
ARCH = "gfx950"


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 into one shared library."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvsde_hip.so")
    # -fno-slp-vectorize: on gfx950 a wave64 v_pk_fma_f32 / v_pk_add_f32 issues in 8 cycles (no gain over two scalar ops) and the
    # SLP vectorizer pays extra v_mov's to build the packed operands of the GRU step loops (measured: -7 % VALU issue cycles)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC",
           "-o", LIB_PATH] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
