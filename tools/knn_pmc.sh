#!/bin/bash
# PMC passes over one screened + one exact kNN call.  bash tools/knn_pmc.sh [D]
D=${1:-64}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/knnpmc
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $OUT/p$i -o p --output-format csv -- python3 tools/knn_pmc.py $D > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "knn_screen" in n or "knn_mfma" in n:
            key = ("screen" if "screen" in n else "exact") + " grid=" + r.get("Grid_Size", "?")
            acc[key][r["Counter_Name"]] = acc[key].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k, v in acc.items():
    print(k)
    for c, x in v.items():
        print(f"   {c:32s} {x:16.0f}")
PY
rm -f $OUT/p*/p_counter_collection.csv
