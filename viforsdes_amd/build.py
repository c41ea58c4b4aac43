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
SOURCES = ["vsde_gemm.hip", "vsde_head.hip", "vsde_elbo.hip", "vsde_encoder.hip", "vsde_wgrad.hip", "vsde_attn.hip", "vsde_sde.hip", "vsde_linear.hip", "vsde_attn_stream.hip", "vsde_tn_wide.hip", "vsde_proj.hip", "vsde_pack.hip", "vsde_optim.hip", "vsde_head_mp.hip", "vsde_mlp.hip"]
HEADERS = ["vsde_common.h", os.path.join("..", "..", "include", "vsde_hip.h")]
ARCH = "gfx950"


def _stale(lib_path: str = LIB_PATH) -> bool:
    if not os.path.exists(lib_path):
        return True
    t = os.path.getmtime(lib_path)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


ABL_LIB_PATH = os.path.join(_PKG, "libvsde_hip_abl.so")


def build_library(force: bool = False, verbose: bool = False, ablations: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 (one object per source, in parallel, rebuilt only when the source or a
    header changed) and link them into one shared library.  ``ablations``: the tools' build (-DVSDE_ABLATIONS ->
    libvsde_hip_abl.so: A/B switches read from the environment, losing kernel variants compiled in; load it with VSDE_HIP_LIB)."""
    if ablations:
        if not force and not _stale(ABL_LIB_PATH):
            return ABL_LIB_PATH
        return _build(force, verbose, ABL_LIB_PATH, ".obj_abl", ["-DVSDE_ABLATIONS"])
    if not force and not _stale():
        return LIB_PATH
    return _build(force, verbose, LIB_PATH, ".obj", [])


def _build(force: bool, verbose: bool, lib_path: str, objname: str, extra: list) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvsde_hip.so")
    objdir = os.path.join(CSRC, objname)
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    # -fno-slp-vectorize: on gfx950 a wave64 v_pk_fma_f32 / v_pk_add_f32 issues in 8 cycles (no gain over two scalar ops) and the
    # SLP vectorizer pays extra v_mov's to build the packed operands of the GRU step loops (measured: -7 % VALU issue cycles)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC"] + extra

    def compile_one(src: str):
        path, obj = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            return obj, None
        cmd = [hipcc] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        return obj, (None if res.returncode == 0 else f"{src}:\n{res.stdout}\n{res.stderr}")

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    errors = [e for _, e in results if e]
    if errors:
        raise RuntimeError("hipcc failed:\n" + "\n".join(errors))
    # -s: no host-side static symbol table (the C-ABI exports are dynamic symbols; kernel names live in the embedded code object)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-s", "-o", lib_path] + [o for o, _ in results]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{res.stdout}\n{res.stderr}")
    return lib_path


if __name__ == "__main__":
    _argv = __import__("sys").argv
    print(build_library(force="--force" in _argv, verbose=True, ablations="--ablations" in _argv))
