#!/bin/bash
# All summaries committed under profiles/ for the round: kernel statistics of every bench configuration, the captured
# step, PMC traffic / MFMA utilisation, the gemm_nt and gemm_tn shape tables.  bash tools/final_prof.sh
mkdir -p gpurun_out/round
bash tools/profile_round.sh stats > gpurun_out/round_stats.log 2>&1
bash tools/profile_round.sh graph >> gpurun_out/round_stats.log 2>&1
bash tools/profile_round.sh pmc >> gpurun_out/round_stats.log 2>&1
PCB_TN_TRACE=1 PCB_NT_TRACE=1 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --exec eager 2> gpurun_out/round/tn_trace.txt > /dev/null
python tools/tn_bench.py gpurun_out/round/tn_trace.txt > gpurun_out/round/tn_bench.log 2>&1
python tools/nt_bench.py gpurun_out/round/tn_trace.txt > gpurun_out/round/nt_bench_trace.log 2>&1
python tools/fused_bench.py gpurun_out/round/tn_trace.txt > gpurun_out/round/fused_bench.log 2>&1
python tools/knn_screen_bench.py > gpurun_out/round/knn_screen_bench.log 2>&1
for n in pn2_msg_bf16 pn2_msg_bf16_single_stream pn2_msg_fp32 pn2_msg_bf16_infer dgcnn_bf16 bridgeseg_bf16 pn2_msg_bf16_graph; do
  python tools/prof_categories.py gpurun_out/round/$n 0 > gpurun_out/round/${n}_categories.txt 2>&1
done
rm -rf gpurun_out/round/*/p_kernel_trace.csv gpurun_out/round/pmc_*/p*_counter_collection.csv
tail -3 gpurun_out/round_stats.log; tail -1 gpurun_out/round/nt_bench.log; tail -1 gpurun_out/round/tn_bench.log
