#!/bin/bash
# rocprofv3 kernel trace of the headline bench (13 steps incl. warm-up) -> gpurun_out/<name>/ ; then
#   python tools/prof_categories.py gpurun_out/<name> 13
# usage (on the GPU box): bash tools/prof_step.sh <name> [extra bench.py flags]
set -e
NAME=$1; shift
REPO=$PWD
mkdir -p $REPO/gpurun_out/$NAME
cd /tmp && export TMPDIR=/tmp
PCB_BENCH_NO_ROOFLINE=1 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/$NAME -o p --output-format csv -- \
    python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --exec eager "$@" > $REPO/gpurun_out/$NAME/bench.json 2> $REPO/gpurun_out/$NAME/err.log
cd $REPO
ls gpurun_out/$NAME
