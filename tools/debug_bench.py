import sys, time, faulthandler, torch
faulthandler.dump_traceback_later(100, repeat=True, file=sys.stderr)
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from pointcloud_bridge_amd import ops
t0 = time.time()
def log(m):
    torch.cuda.synchronize(); print(f"[{time.time()-t0:7.2f}s] {m}", flush=True)
B, N = int(sys.argv[1]), int(sys.argv[2])
model, cdim = bench.build_model(sys.argv[3] if len(sys.argv) > 3 else "pn2_msg")
model = model.cuda().train()
xyz, colors, labels = bench.synthetic_batch(B, N, 0, "cuda")
log("data ready")
hooks = []
for name, m in model.named_children():
    m.register_forward_hook(lambda mod, i, o, name=name: log(f"fwd {name}"))
for it in range(3):
    out = model(xyz, colors); log("forward done")
    loss = bench.loss_fn(out, labels, cdim); log("loss")
    loss.backward(); log("backward done")
