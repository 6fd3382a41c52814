#!/bin/bash
# Ordered kernel timeline of one REPLAYED step of a bench model: bash tools/tl_graph_model.sh <name> <model> [bench flags]
NAME=${1:-tl_graph}; MODEL=${2:-pn2_msg}; shift 2; R=$PWD; O=$R/gpurun_out/$NAME; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PCB_BENCH_NO_ROOFLINE=1 rocprofv3 --kernel-trace -d $O/tl -o p --output-format csv -- python3 $R/bench.py --model $MODEL --steps 4 --warmup 3 --no-cpu-baseline --no-extras --exec graph "$@" > $O/tl_bench.json 2> $O/tl_err.log
cd $R; python tools/timeline.py $O/tl 2 > $O/timeline.txt; rm -rf $O/tl; tail -3 $O/timeline.txt
