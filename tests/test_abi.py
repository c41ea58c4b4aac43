"""CPU: the C-ABI library exists, loads and exports every symbol declared in include/vsde_hip.h;
the product path refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "vsde_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vsde_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from viforsdes_amd.build import build_library
    lib = ctypes.CDLL(build_library())
    names = _declared_functions()
    assert "vsde_head_forward" in names and "vsde_head_backward" in names and "vsde_elbo_path_terms" in names
    for n in names:
        assert hasattr(lib, n), f"libvsde_hip.so does not export {n}"
    lib.vsde_abi_version.restype = ctypes.c_int
    assert lib.vsde_abi_version() == 1


def test_python_binding_lists_the_same_symbols():
    from viforsdes_amd import _hip
    assert sorted(_hip.EXPORTS) == _declared_functions()
    _hip.load()


def test_workspace_queries_and_argument_errors_without_gpu():
    from viforsdes_amd import _hip
    lib = _hip.load()
    d = _hip._Dims(512, 400, 2, 3, 256, 64, 2)
    assert lib.vsde_head_forward_workspace_bytes(ctypes.byref(d)) >= 512 * 400 * 192 * 4
    assert lib.vsde_head_backward_workspace_bytes(ctypes.byref(d)) >= 512 * 400 * 2 * 256 * 4
    bad = _hip._Dims(8, 10, 2, 3, 16, 64, 5)  # 5 layers
    assert lib.vsde_head_forward_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"num_layers" in lib.vsde_last_error()


def test_cpu_tensors_are_refused_not_emulated():
    from viforsdes_amd import HeadConfig, _hip
    from viforsdes_amd.models.head import DiffusionTransitionHead
    head = DiffusionTransitionHead(2, 8, 3, HeadConfig(hidden_dim=8, num_layers=1))
    with pytest.raises(_hip.HipLibraryError):
        head.sample_diffusion_paths(torch.zeros(2, 2), torch.zeros(2, 4, 8), torch.ones(2, 3), torch.zeros(2, 4, 2), 0.1)
    with pytest.raises(ValueError):
        DiffusionTransitionHead(2, 8, 3, HeadConfig(hidden_dim=8, num_layers=5))
