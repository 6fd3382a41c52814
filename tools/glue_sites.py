"""Which lines of the package issue the ATen glue ops (copies, fills, adds, cats) of one training step?
A TorchDispatchMode logs every aten op with the innermost frame inside this repo -- forward pass and,
by running the backward pass on the calling thread is not possible, the Python-level backward
functions only (C++ autograd nodes have no Python frame and are listed under '<autograd engine>')."""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointcloud_bridge_amd import parallel, rowmlp
from torch.utils._python_dispatch import TorchDispatchMode
rowmlp.set_precision("bf16")
torch.manual_seed(42)
model, cdim = bench.build_model(sys.argv[1] if len(sys.argv) > 1 else "pn2_msg")
model = model.cuda().train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = parallel.FlatGradAllReduce(params, assign_views=False)
opt = parallel.FlatAdam(params, lr=1e-3, weight_decay=1e-4)
xyz, colors, labels = bench.synthetic_batch(16 if cdim == 1 else 8, 16384 if cdim == 1 else 8192, 0, "cuda")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WATCH = ("copy_", "_to_copy", "fill_", "zero_", "zeros", "add", "cat", "clone", "constant_pad_nd", "contiguous", "sum", "mul", "index")
sites = collections.Counter()
volume = collections.Counter()

class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if any(name == w or name.startswith(w) for w in WATCH):
            frame = "<autograd engine>"
            for fs in reversed(traceback.extract_stack()[:-1]):
                if root in fs.filename and "tools/" not in fs.filename:
                    frame = f"{fs.filename.replace(root + '/', '')}:{fs.lineno} {fs.line}"
                    break
            out = func(*args, **(kwargs or {}))
            first = out if torch.is_tensor(out) else (args[0] if args and torch.is_tensor(args[0]) else None)
            sites[(name, frame)] += 1
            volume[(name, frame)] += 0 if first is None else first.numel() * first.element_size()
            return out
        return func(*args, **(kwargs or {}))

def step():
    bucket.zero(); loss = bench.loss_fn(model(xyz, colors), labels, cdim); loss.backward(); opt.step(None)
for _ in range(2): step()
with Log():
    step()
torch.cuda.synchronize()
for (name, frame), n in sorted(sites.items(), key=lambda kv: (-volume[kv[0]], kv[0]))[:40]:
    print(f"{volume[(name, frame)] / 1e6:8.1f} MB {n:4d}x {name:18s} {frame[:140]}")
