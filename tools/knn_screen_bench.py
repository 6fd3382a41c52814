"""Scratch: screened (split-bf16 MFMA + exact recheck) against the exact all-pairs kNN kernel: time, recomputed queries."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import ops

torch.manual_seed(0)
B, N = int(os.environ.get("KNN_B", "8")), int(os.environ.get("KNN_N", "8192"))


def clouds(D):
    g = torch.randn(B, N, D, device="cuda")
    yield "gaussian", g
    # post BatchNorm + LeakyReLU(0.2) features of a clustered cloud (what DGCNN's later graphs see)
    centres = torch.randn(B, 64, D, device="cuda") * 2
    lab = torch.randint(0, 64, (B, N), device="cuda")
    f = torch.gather(centres, 1, lab.unsqueeze(-1).expand(B, N, D)) + 0.3 * g
    f = (f - f.mean(dim=(0, 1))) / f.std(dim=(0, 1))
    yield "clustered bn+lrelu", torch.nn.functional.leaky_relu(f, 0.2).contiguous()
    yield "offset +20", (g + 20).contiguous()


for D in (64, 128):
    for name, x in clouds(D):
        res = {}
        for flag in (False, True):
            ops.set_screen_knn(flag)
            for _ in range(3): out = ops.knn(x, 20)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): out = ops.knn(x, 20)
            b.record(); torch.cuda.synchronize()
            res[flag] = (a.elapsed_time(b) / 10 * 1e3, out)
        ops.collect_knn_stats(True)
        ops.knn(x, 20)
        st = ops.collect_knn_stats(False)
        rec = int(st[0][4].sum())
        same = torch.equal(res[False][1], res[True][1])
        print(f"D={D} {name:20s} exact {res[False][0]:6.0f} us  screened {res[True][0]:6.0f} us  recomputed {rec}/{B*N}  identical {same}", flush=True)
