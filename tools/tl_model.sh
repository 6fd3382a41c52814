#!/bin/bash
# Ordered single-stream kernel timeline of one step of a bench model: bash tools/tl_model.sh <name> <bench flags...>
NAME=$1; shift; R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PCB_BRANCH_STREAMS=0 PCB_BENCH_NO_ROOFLINE=1 rocprofv3 --kernel-trace -d $O/tl -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras --exec eager "$@" > $O/tl_bench.json 2> $O/tl_err.log
cd $R; python tools/timeline.py $O/tl 2 > $O/timeline.txt; rm -rf $O/tl; tail -3 $O/timeline.txt
