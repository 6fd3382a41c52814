import argparse, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from pointcloud_bridge_amd.models import pointnet2_utils as pu
dev = torch.device("cuda", 0)
def flat_of(segments):
    args = argparse.Namespace(no_dropout=True, no_prefetch=False, dump=False, graph_segments=segments)
    run = bench.Run(args, "pn2_msg", "bf16", 4, 4096, 0, 1, dev, graph=True)
    run.opt.step = lambda *a, **k: None
    out = []
    for name, step in (("eager", run.eager_step), ("graph", run.graph_step), ("graph", run.graph_step), ("eager", run.eager_step)):
        run.i = 0
        torch.manual_seed(11)
        step()
        loss = step()
        torch.cuda.synchronize()
        out.append((name, float(loss), run.bucket.flat.clone()))
    run.close(); pu.set_static_sampling(None)
    return out
one, two = flat_of(1), flat_of(2)
for tag, res in (("one", one), ("two", two)):
    print(tag, [(n, round(l, 5), float(f.norm())) for n, l, f in res])
    for i in range(len(res)):
        for j in range(i + 1, len(res)):
            print("   ", tag, i, j, float((res[i][2] - res[j][2]).norm() / res[i][2].norm()))
for i in range(4):
    print("one vs two", i, float((one[i][2] - two[i][2]).norm() / one[i][2].norm()))
# per top-level module difference between the one-graph and two-segment flats (graph replay #1)
from pointcloud_bridge_amd.models.containers import PointNet2MSG
off = 0
a, b = one[1][2], two[1][2]
mods = {}
for name, prm in PointNet2MSG(5).named_parameters():
    n = prm.numel()
    top = name.split(".")[0]
    d = mods.setdefault(top, [0.0, 0.0])
    d[0] += float((a[off:off + n] - b[off:off + n]).pow(2).sum()); d[1] += float(a[off:off + n].pow(2).sum())
    off += n
print({k: round((v[0] / max(v[1], 1e-30)) ** 0.5, 4) for k, v in mods.items()})
