#!/bin/bash
# kernel time of the screen kernel under experimental builds (exp_build/libpcb_*.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf gpurun_out/knnexp_$v
  if [ "$v" = base ]; then L=""; else L=$PWD/exp_build/libpcb_$v.so; fi
  PCB_LIB=$L timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/knnexp_$v -o knn --output-format csv -- python3 tools/knn_pmc.py 64 > gpurun_out/knnexp_$v.log 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/knnexp_$v/knn_kernel_stats.csv")):
    if "knn_screen" in r["Name"]: print("$v", "screen kernel", r["AverageNs"])
PY
  rm -f gpurun_out/knnexp_$v/knn_kernel_trace.csv
done
