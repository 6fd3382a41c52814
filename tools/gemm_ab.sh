#!/bin/bash
# nt + tn shape lists (totals) and the bf16 / fp32 kernel tests: bash tools/gemm_ab.sh <name>
O=gpurun_out/${1:-gemm_ab}; mkdir -p $O
python tools/nt_bench.py > $O/nt.log 2>&1; tail -1 $O/nt.log
PCB_TN_TRACE=1 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --exec eager 2> $O/trace.txt > /dev/null
python tools/tn_bench.py $O/trace.txt > $O/tn.log 2>&1; tail -1 $O/tn.log
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_round2.py tests/test_gpu_modules.py -x -q 2>&1 | tail -2
