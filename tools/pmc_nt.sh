#!/bin/bash
# PMC passes over a few gemm_nt shapes (tools/nt_bench.py --only ...): SQ activity / wait counters.
# usage (on the GPU box): bash tools/pmc_nt.sh "4,30,31,43,69" outdir [launch trace for nt_bench.py, relative to the repo]
set -e
ONLY=$1; OUT=$2; REPO=$PWD; TRACE=${3:+$REPO/$3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
  -d $REPO/$OUT/p1 -o p1 --output-format csv -- python3 $REPO/tools/nt_bench.py $TRACE --only $ONLY --reps 1 > $REPO/$OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVES \
  -d $REPO/$OUT/p2 -o p2 --output-format csv -- python3 $REPO/tools/nt_bench.py $TRACE --only $ONLY --reps 1 > $REPO/$OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM \
  -d $REPO/$OUT/p3 -o p3 --output-format csv -- python3 $REPO/tools/nt_bench.py $TRACE --only $ONLY --reps 1 > $REPO/$OUT/p3.log 2>&1
