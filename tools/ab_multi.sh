#!/bin/bash
# Same-box comparison of several values of one environment knob: bash tools/ab_multi.sh VAR "v1 v2 v3" [rounds] [bench flags...]
VAR=$1; VALS=$2; N=${3:-2}; shift 3 2>/dev/null; mkdir -p gpurun_out/ab
for i in $(seq $N); do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-extras --no-cpu-baseline --steps 40 --exec graph "$@" > gpurun_out/ab/${VAR}_${v}_$i.json 2> gpurun_out/ab/err.log
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/${VAR}_${v}_$i.json").read().strip().splitlines()[-1])
print("$VAR=$v run $i: %.3f ms/step" % d["ms_per_step"])
PY
  done
done
