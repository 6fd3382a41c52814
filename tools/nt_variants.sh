#!/bin/bash
# gemm_nt shape list under the experiment knobs of launch_nt (csrc/gemm.hip): bash tools/nt_variants.sh
mkdir -p gpurun_out/ntv
python tools/nt_bench.py > gpurun_out/ntv/base.log 2>&1; tail -1 gpurun_out/ntv/base.log
PCB_NT8_PLAIN_ANYK=1 python tools/nt_bench.py > gpurun_out/ntv/anyk.log 2>&1; tail -1 gpurun_out/ntv/anyk.log
PCB_NT8_PLAIN_ANYK=1 PCB_NT8_SMALLR=16384 python tools/nt_bench.py > gpurun_out/ntv/small16k.log 2>&1; tail -1 gpurun_out/ntv/small16k.log
PCB_NT8_PLAIN_ANYK=1 PCB_NT8_SMALLR=65536 python tools/nt_bench.py > gpurun_out/ntv/small64k.log 2>&1; tail -1 gpurun_out/ntv/small64k.log
