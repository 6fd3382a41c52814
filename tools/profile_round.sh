#!/bin/bash
# The rocprofv3 summaries committed under profiles/ for one round (run on the GPU box):
#   bash tools/profile_round.sh stats   -> kernel-trace statistics of the bench configurations
#   bash tools/profile_round.sh pmc     -> HBM traffic and MFMA utilisation of the GEMM kernels, gemm_nt shape table
# Output: gpurun_out/round/<name>...; copy what is to be judged into profiles/ (see profiles/README.md).
set -e
REPO=$PWD
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, extra env as VAR=VALUE or "-", bench flags...
    local name=$1 envv=$2; shift 2
    [ "$envv" != "-" ] && export $envv
    PCB_BENCH_NO_ROOFLINE=1 rocprofv3 --kernel-trace --stats -d $OUT/$name -o p --output-format csv -- \
        python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --exec eager "$@" > $OUT/$name.json 2> $OUT/$name.err
    [ "$envv" != "-" ] && unset ${envv%%=*}
    cp $OUT/$name/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv
    echo "$name done"
}
pmc() {  # name, counters...
    local name=$1; shift
    PCB_BRANCH_STREAMS=0 rocprofv3 --pmc "$@" -d $OUT/$name -o p --output-format csv -- \
        python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --exec eager --no-prefetch > $OUT/$name.json 2> $OUT/$name.err
    echo "$name done"
}
case "$1" in
stats)
    stats pn2_msg_bf16 -
    stats pn2_msg_bf16_single_stream PCB_BRANCH_STREAMS=0
    stats pn2_msg_fp32 - --precision fp32
    stats pn2_msg_bf16_infer - --mode infer
    stats dgcnn_bf16 - --model dgcnn
    stats bridgeseg_bf16 - --model bridgeseg
    stats ptv3_bf16_infer - --model ptv3 --mode infer
    ;;
graph)
    # the captured step: bench.py --exec graph (the stats helper passes --exec eager first; the later flag wins)
    stats pn2_msg_bf16_graph - --exec graph
    ;;
pmc)
    pmc pmc_fetch FETCH_SIZE
    pmc pmc_write WRITE_SIZE
    pmc pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES
    CMD="PCB_BRANCH_STREAMS=0 rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --exec eager --no-prefetch"
    python3 $REPO/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_gemm_nt_bf16.json "$CMD"
    python3 $REPO/tools/mfma_util.py $OUT/pmc_mfma $OUT/pmc_mfma_util.json "$CMD"
    cd $REPO && python3 tools/nt_bench.py > $OUT/nt_bench.log 2>&1
    ;;
esac
ls $OUT | head -40
