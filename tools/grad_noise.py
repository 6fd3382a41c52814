"""Noise floor of the first-step gradient: two identical single-process bench runs (same seeds, same
data) compared with each other -- the yardstick for the two-rank equivalence test."""
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
common = ["--steps", "1", "--warmup", "0", "--npoints", "2048", "--batch", "4", "--precision", sys.argv[1] if len(sys.argv) > 1 else "fp32",
          "--no-cpu-baseline", "--no-extras", "--no-dropout", "--model", "pn2_msg", "--gpus", "1"]
outs = []
for i in range(2):
    path = f"/tmp/noise{i}.pt"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dump", path] + common, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    outs.append(torch.load(path, weights_only=True))
a, b = outs[0]["first_grad"], outs[1]["first_grad"]
print("two identical single-process runs: max |d| / max |g| =", float((a - b).abs().max() / a.abs().max()),
      " losses", outs[0]["losses"], outs[1]["losses"])
