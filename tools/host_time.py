"""Host enqueue time of one bench training step with an EMPTY GPU queue at its start (synchronise, then time the
Python call): how long the host needs by itself, which is what bounds the step once the GPU is faster.
Also: cProfile of that call, sorted by own time (GPU waits cannot pollute it: the queue never fills in one step).

    python tools/host_time.py [model]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("model", nargs="?", default="pn2_msg")
a = ap.parse_args()
args = argparse.Namespace(no_dropout=False, no_prefetch=False, dump=False)
dev = torch.device("cuda", 0)
B, N = (8, 8192) if a.model in ("dgcnn",) else (16, 16384)
run = bench.Run(args, a.model, "bf16", B, N, 0, 1, dev)
for _ in range(6):
    run.train_step()
import gc
gc.collect(); gc.disable()
ts, tg = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.train_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(t1 - t0); tg.append(t2 - t0)
print(f"host enqueue {min(ts) * 1e3:.2f} ms (median {sorted(ts)[5] * 1e3:.2f}); enqueue + drain {min(tg) * 1e3:.2f} ms")
pr = cProfile.Profile()
for _ in range(5):
    torch.cuda.synchronize()
    pr.enable()
    run.train_step()
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(40)
