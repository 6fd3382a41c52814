#!/bin/bash
# Same-box A/B of an environment knob on a bench configuration: bash tools/ab_env.sh VAR a b [rounds] [bench flags...]
# (alternating runs; default = the headline bench)
VAR=$1; A=$2; B=$3; N=${4:-2}; shift 4 2>/dev/null; mkdir -p gpurun_out/ab
for i in $(seq $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --no-extras --no-cpu-baseline --steps 40 "$@" > gpurun_out/ab/${VAR}_${v}_$i.json 2> gpurun_out/ab/err.log
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/${VAR}_${v}_$i.json").read().strip().splitlines()[-1])
print("$VAR=$v run $i: %.3f ms/step (%s), eager %.3f graph %.3f" % (d["ms_per_step"], d["config"]["exec"]["mode"], d["config"]["exec"].get("eager_ms",0), d["config"]["exec"].get("graph_ms",0)))
PY
  done
done
