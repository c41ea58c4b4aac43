"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ablation_build: exercises kernel variants / switches that only exist in the tools' library "
                            "(libvsde_hip_abl.so, built with -DVSDE_ABLATIONS): the test runs with that library swapped in")


@pytest.fixture(autouse=True)
def _ablation_library(request, monkeypatch):
    """Tests marked ``ablation_build`` run on libvsde_hip_abl.so: measured-and-losing kernel variants and the environment switches that
    select them are not compiled into the shipped library (csrc/vsde_common.h).  The ctypes handle of ``viforsdes_amd._hip`` is swapped
    for the test and VSDE_HIP_LIB is set for the child processes some of them start."""
    if request.node.get_closest_marker("ablation_build") is None:
        yield
        return
    import ctypes
    from viforsdes_amd import _hip
    from viforsdes_amd.build import build_library
    path = build_library(ablations=True)
    shipped, shipped_path = _hip.load(), _hip._loaded_path
    lib = ctypes.CDLL(path)
    _hip._lib = None
    monkeypatch.setenv("VSDE_HIP_LIB", path)
    try:
        _hip.load()
        assert _hip.has_ablations()
        yield
    finally:
        _hip._lib, _hip._loaded_path = shipped, shipped_path


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
