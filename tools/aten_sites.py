"""Scratch: the torch-level calls (ATen launches) of one step of the captured pn2_msg configuration run eagerly, with the
Python line of this repo that makes each of them (TorchFunctionMode; the autograd engine's own gradient accumulation is
not a Python call and does not appear).  python tools/aten_sites.py [model]"""
import os, sys, collections, types, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.overrides import TorchFunctionMode

model = sys.argv[1] if len(sys.argv) > 1 else "pn2_msg"
args = types.SimpleNamespace(no_dropout=False, no_prefetch=False, dump=None, graph_segments=1, dup_halves=False)
from pointcloud_bridge_amd import parallel
parallel.init_from_env()
dev = torch.device("cuda", 0)
run = bench.Run(args, model, "bf16", (8 if model == "dgcnn" else 16), (8192 if model == "dgcnn" else 16384), 0, 1, dev, "train", "ball", True, "ce", False, False)
for _ in range(3): run.eager_step()
torch.cuda.synchronize()
SKIP = {"size", "dim", "view", "reshape", "data_ptr", "is_contiguous", "stride", "__get__", "numel", "expand", "transpose",
        "permute", "unsqueeze", "squeeze", "__getitem__", "shape", "device", "dtype", "detach", "requires_grad_", "is_cuda",
        "element_size", "storage_offset", "view_as", "unbind", "split", "chunk", "narrow", "select", "t", "flatten", "unflatten",
        "record_stream", "untyped_storage", "_is_view", "apply", "type", "get_device", "is_floating_point", "__len__", "__set__",
        "dim_order", "as_strided", "nelement", "ndimension", "__bool__", "item", "is_pinned", "backward", "retain_grad"}
log = collections.Counter()

class Spy(TorchFunctionMode):
    def __torch_function__(self, func, types_, a=(), kw=None):
        out = func(*a, **(kw or {}))
        name = getattr(func, "__name__", str(func))
        if name in SKIP: return out
        same = isinstance(out, torch.Tensor) and any(isinstance(x, torch.Tensor) and x.data_ptr() == out.data_ptr() and x.dtype == out.dtype
                                                     and x.shape == out.shape for x in a) and name in ("contiguous", "to", "float", "bfloat16", "clone")
        if same and name != "clone": return out          # no-op conversions launch nothing
        fr = [f for f in traceback.extract_stack()[:-1] if ("bridge_amd" in f.filename or f.filename.endswith("bench.py"))]
        site = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:][::-1]) if fr else "?"
        shape = tuple(out.shape) if isinstance(out, torch.Tensor) else ""
        log[(name, site, str(shape), str(getattr(out, "dtype", "")).replace("torch.", ""))] += 1
        return out

import importlib, inspect
def wrap(fn):
    def inner(*a, **k):
        with Spy():
            return fn(*a, **k)
    return inner
for mn in ("rowmlp", "ops", "rowsf32", "losses", "models.pointnet2_utils", "models.containers"):
    mod = importlib.import_module("pointcloud_bridge_amd." + mn)
    for _, cls in inspect.getmembers(mod, inspect.isclass):
        if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and "backward" in cls.__dict__:
            cls.backward = staticmethod(wrap(cls.__dict__["backward"].__func__))
with Spy():
    run.eager_step()
torch.cuda.synchronize()
for (name, site, shape, dt), n in sorted(log.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{n:3d} {name:18s} {shape:24s} {dt:9s} {site}")
print(sum(log.values()), "calls")
