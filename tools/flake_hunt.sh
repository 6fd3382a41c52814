#!/bin/bash
# The GPU suite N times; prints every failing test id (a flaky test must not reach the round-end run)
N=${1:-6}; mkdir -p gpurun_out/flake
for i in $(seq 1 $N); do
  python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/flake/run$i.log 2>&1
  grep -E "^FAILED|passed|failed" gpurun_out/flake/run$i.log | tail -3
done
