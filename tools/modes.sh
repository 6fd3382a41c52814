#!/bin/bash
# The headline bench in each execution mode on ONE box (host speed differs from box to box): usage bash tools/modes.sh <name> [modes...]
NAME=${1:-modes}; shift; O=gpurun_out/$NAME; mkdir -p $O
for m in ${@:-auto graph eager}; do
  python bench.py --no-extras --no-cpu-baseline --exec $m > $O/b_$m.json 2> $O/b_$m.err
  python - <<PY
import json
d = json.load(open("$O/b_$m.json"))
print("$m", round(d["ms_per_step"], 3), "host", round(d["config"]["host_enqueue_ms_per_step"], 2), d["config"]["exec"])
PY
done
