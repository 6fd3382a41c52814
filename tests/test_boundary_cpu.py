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


def _desc(widths, Kp, gathered=False):
    """Host descriptor of pcb_mlp_stack_* (PCB_STACK_DESC_SLOTS = 18 int64 per layer) with dummy non-null pointers."""
    vals, kp = [], Kp
    for l, C in enumerate(widths):
        w = 0 if (gathered and l == 0) else 64
        vals += [w, 0, 0, 0, 0, 0, C, 0 if w == 0 else kp, 1, 64, 0, 0, 0, 0, 0, 0, 0, 0]
        kp = C
    return (ctypes.c_longlong * len(vals))(*vals)


def test_operand_table_layout_mirrors_the_stack_runtime(lib):
    """rowmlp.prepare_step fills every stack's operand buffer from a table built on the host: the offsets it
    computes (rowmlp._step_rows) must be the layout csrc/stack.hip's parse() uses inside wbuf -- checked through the
    one number the library exposes, pcb_mlp_stack_wbuf_elems, and the no-overlap / in-order structure of the rows."""
    import torch.nn as nn
    from pointcloud_bridge_amd import rowmlp
    L = lib.load()
    for widths, Kp, need in (([64, 64, 128], 8, 0), ([64, 64, 128], 8, 1), ([1024, 256], 1536, 1), ([384], 264, 0), ([16], 8, 1)):
        ws, kin = [], Kp
        for C in widths:
            ws.append(nn.Parameter(torch.zeros(C, kin if ws else max(Kp - 2, 1))))   # layer 0: real k < padded Kp
            kin = C
        rows, total = rowmlp._step_rows(ws, Kp, 0, need, 8)
        assert total == L.pcb_mlp_stack_wbuf_elems(len(widths), _desc(widths, Kp), Kp, need)
        end = 0
        for (l, C, k, kp, perm, wp, wt) in rows:
            assert wp == end and kp == (Kp if l == 0 else widths[l - 1]) and k == ws[l].shape[1]
            end = wp + C * kp
            if l > 0 or need:
                assert wt == end
                end += C * kp
            else:
                assert wt == -1
        assert end == total
    # gathered first layer: no weights of its own, the following layers read rows of its width
    ws = [None, nn.Parameter(torch.zeros(64, 64)), nn.Parameter(torch.zeros(128, 64))]
    rows, total = rowmlp._step_rows(ws, 64, 0, False, 8)
    assert total == L.pcb_mlp_stack_wbuf_elems(3, _desc([64, 64, 128], 0, gathered=True), 0, 0) == 2 * (64 * 64 + 128 * 64)


def test_gradient_scratch_sizing(lib):
    """pcb_mlp_stack_dzbuf_elems: two ping-pong slots of the widest inner gradient; a top layer whose dy is written
    out once (bf16 rows, C > 256, more than 256 inputs, no pooling) widens them / makes them exist for one layer."""
    L = lib.load()
    R = 1000
    assert L.pcb_mlp_stack_dzbuf_elems(0, 3, _desc([64, 64, 128], 8), R, 8, 0, 0) == 2 * R * 64
    assert L.pcb_mlp_stack_dzbuf_elems(0, 1, _desc([128], 264), R, 264, 0, 0) == 0
    assert L.pcb_mlp_stack_dzbuf_elems(0, 1, _desc([384], 264), R, 264, 0, 0) == 2 * R * 384      # dy slot for the single layer
    assert L.pcb_mlp_stack_dzbuf_elems(0, 2, _desc([512, 320], 384), R, 384, 0, 0) == 2 * R * 512  # inner 512 (in place), top 320
    assert L.pcb_mlp_stack_dzbuf_elems(0, 2, _desc([256, 512], 384), R, 384, 0, 0) == 2 * R * 384  # top: 256 inputs only -> not written out
    assert L.pcb_mlp_stack_dzbuf_elems(0, 2, _desc([320, 512], 384), R, 384, 0, 0) == 2 * R * 512  # top written out
    assert L.pcb_mlp_stack_dzbuf_elems(0, 2, _desc([320, 512], 384), R, 384, 16, 0) == 2 * R * 384  # pooled top: never
    assert L.pcb_mlp_stack_dzbuf_elems(1, 2, _desc([320, 512], 384), R, 384, 0, 0) == 2 * R * 384   # fp32 rows: never


def test_cross_entropy_on_cpu_is_the_reference_call(lib):
    """losses.cross_entropy: CPU logits (the harness tests, the CPU port) take F.cross_entropy, the trainers' own call."""
    import torch.nn.functional as F
    from pointcloud_bridge_amd.losses import cross_entropy
    torch.manual_seed(0)
    logits = torch.randn(2, 5, 64, requires_grad=True)
    labels = torch.randint(0, 5, (2, 64))
    labels[0, :3] = -100
    a = cross_entropy(logits, labels)
    assert torch.equal(a, F.cross_entropy(logits, labels))
    b = cross_entropy(logits.transpose(1, 2), labels, channels_last=True)
    assert torch.allclose(a, b)


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


REF = "/root/reference/Highway_bridge"
_DUMP = r'''
import hashlib, json, sys
import torch
REPO, REF, dropin = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
import models                                   # the reference's package (empty __init__)
if dropin:                                      # what INTEGRATION.md section 2 tells a maintainer to do
    from pointcloud_bridge_amd.models import pointnet2_utils as drop
    sys.modules["models.pointnet2_utils"] = drop
    models.pointnet2_utils = drop
from models.model import EnhancedPointNet2, PointNet2          # the reference's containers, unchanged
from models.pointnet2 import PointNet2 as PointNet2Skip
out = {}
for name, ctor in (("ssg", lambda: PointNet2(5)), ("ssg_skip", lambda: PointNet2Skip(5)), ("bridgeseg", lambda: EnhancedPointNet2(5))):
    torch.manual_seed(42)
    sd = ctor().state_dict()
    out[name] = [(k, list(v.shape), hashlib.sha1(v.detach().cpu().numpy().tobytes()).hexdigest()) for k, v in sd.items()]
if dropin:
    import models.model as mm
    assert mm.SetAbstraction.__module__.startswith("pointcloud_bridge_amd"), mm.SetAbstraction.__module__
print(json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_reference_containers_build_unchanged_on_the_dropin(lib):
    """INTEGRATION.md section 2, exercised: the reference's own containers (models/model.py:12-147,
    models/pointnet2.py:10-61) imported over the drop-in `pointnet2_utils` construct the same networks --
    same state_dict keys, shapes and (same seed) parameter values -- as over the reference's module."""
    import json
    import subprocess
    import sys
    res = []
    for flag in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", _DUMP, REPO, REF, flag], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        assert r.returncode == 0, r.stderr[-3000:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    ref, mine = res
    for name in ref:
        assert [k for k, _, _ in mine[name]] == [k for k, _, _ in ref[name]], name
        assert mine[name] == ref[name], f"{name}: parameter shapes or seeded values differ"
    assert len(ref["bridgeseg"]) > 300
