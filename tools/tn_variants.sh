#!/bin/bash
# gemm_tn at the shapes of a pn2_msg step (see tools/tn_bench.py); usage on the GPU box: bash tools/tn_variants.sh
mkdir -p gpurun_out/tn
PCB_TN_TRACE=1 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --exec eager 2> gpurun_out/tn/trace.txt > /dev/null
python tools/tn_bench.py gpurun_out/tn/trace.txt > gpurun_out/tn/tn_bench.log 2>&1; tail -1 gpurun_out/tn/tn_bench.log
