#!/bin/bash
# One gpurun call while tuning: GPU tests, the headline bench (no extras), and the ordered kernel timeline of a
# single-stream step.  usage (on the GPU box): bash tools/quick.sh <name> [notest]
NAME=${1:-q}; R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
if [ "$2" != "notest" ]; then
  python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gputest.log
fi
python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("ms/step", round(d["ms_per_step"],3), "host", round(d["config"]["host_enqueue_ms_per_step"],2), "exec", d["config"]["exec"], "frac", round(d["roofline"]["frac"],4), "avg_us", round(d["roofline"]["avg_launch_us"],2), "whole", round(d["roofline"]["whole_step"]["frac"],3))
PY
cd /tmp && export TMPDIR=/tmp
PCB_BRANCH_STREAMS=0 PCB_BENCH_NO_ROOFLINE=1 rocprofv3 --kernel-trace -d $O/tl -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras --exec eager > $O/tl_bench.json 2> $O/tl_err.log
cd $R; python tools/timeline.py $O/tl 2 > $O/timeline.txt; rm -rf $O/tl
tail -3 $O/timeline.txt
