#!/bin/bash
# PMC passes over the flash-attention kernel.  bash tools/attn_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/attnpmc; mkdir -p $OUT; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $OUT/p$i -o p --output-format csv -- python3 tools/attn_pmc.py > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in sorted(glob.glob("$OUT/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "attention_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c, x in acc.items(): print(f"   {c:32s} {x / n[c]:16.0f}   per launch")
PY
rm -f $OUT/p*/p_counter_collection.csv
