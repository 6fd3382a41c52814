"""CPU: the C-ABI library loads, exports every symbol the header declares, validates arguments,
and the product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    from pointcloud_bridge_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        __graft_entry__.build()
    return _lib


def header_symbols():
    text = open(os.path.join(REPO, "include", "pcb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcb_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    syms = header_symbols()
    assert len(syms) >= 16
    so = ctypes.CDLL(lib.LIB_PATH)
    for s in syms:
        assert hasattr(so, s), f"{s} declared in include/pcb_hip.h but not exported"
    assert sorted(lib.SIGNATURES) == syms  # the Python binding covers exactly the declared ABI


def test_version_and_status_strings(lib):
    L = lib.load()
    assert L.pcb_version() == 100
    assert L.pcb_status_string(0) == b"ok"
    assert b"invalid" in L.pcb_status_string(-1)


def test_argument_validation_without_gpu(lib):
    L = lib.load()
    assert L.pcb_fps(None, 1, 8, 4, None, None, None) == -1
    assert L.pcb_ball_query(None, None, 1, 8, 4, 0.1, 4, None, None) == -1
    assert L.pcb_knn(ctypes.c_void_p(8), 1, 8, 3, 99, ctypes.c_void_p(8), ctypes.c_void_p(8), None) == -1      # k > 32
    assert L.pcb_knn(ctypes.c_void_p(8), 1, 64, 400, 4, ctypes.c_void_p(8), ctypes.c_void_p(8), None) == -2    # D > 128
    assert L.pcb_three_nn(ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 8, 2, 3, ctypes.c_void_p(8),
                          ctypes.c_void_p(8), None) == -1                                    # S < k
    with pytest.raises(lib.PcbError):
        lib.check(-2, "x")


def test_ops_refuse_cpu_tensors(lib):
    from pointcloud_bridge_amd import ops
    xyz = torch.rand(1, 16, 3)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.furthest_point_sample(xyz, 4, torch.zeros(1, dtype=torch.long))
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.ball_query(0.1, 4, xyz, xyz[:, :4])
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.knn(xyz, 4)


def test_modules_refuse_cpu_forward_and_keep_reference_names(lib):
    from pointcloud_bridge_amd.models import pointnet2_utils as pu
    for name in ("square_distance", "index_points", "sample_and_group", "farthest_point_sample",
                 "query_ball_point", "SetAbstraction", "FeaturePropagation",
                 "EnhancedFeaturePropagation", "MultiScaleSetAbstraction"):
        assert hasattr(pu, name)
    sa = pu.SetAbstraction(4, 0.5, 4, 6, [8])
    with pytest.raises(RuntimeError, match="GPU only"):
        sa(torch.rand(1, 16, 3), torch.rand(1, 3, 16))


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libpcb_hip.so")
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        lib.load()
