mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for cfg in "512 3" "768 2"; do
    set -- $cfg
    PCB_NT_BWD_GRID=$1 PCB_ARES_WGS=$2 python bench.py --no-extras --no-cpu-baseline --steps 40 > gpurun_out/ab/bwd_$1_$i.json 2> gpurun_out/ab/err.log
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/bwd_$1_$i.json").read().strip().splitlines()[-1])
print("BWD_GRID=$1 ARES_WGS=$2 run $i: %.3f ms/step (%s)" % (d["ms_per_step"], d["config"]["exec"]["mode"]))
PY
  done
done
