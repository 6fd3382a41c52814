#!/bin/bash
# SQ activity / wait counters of gemm_tn at the step's shapes (tools/tn_bench.py under rocprofv3 --pmc): bash tools/pmc_tn.sh
set -e
REPO=$PWD; OUT=$REPO/gpurun_out/pmc_tn; mkdir -p $OUT
PCB_TN_TRACE=1 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --exec eager 2> $OUT/trace.txt > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
  -d $OUT/p1 -o p1 --output-format csv -- python3 $REPO/tools/tn_bench.py $OUT/trace.txt > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU \
  -d $OUT/p2 -o p2 --output-format csv -- python3 $REPO/tools/tn_bench.py $OUT/trace.txt > $OUT/p2.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
for tag in ("p1", "p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_tn_kernel" not in k: continue
        k = k[k.index("gemm_tn_kernel"):][:28]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in sorted(agg):
        print(tag, k, {c: round(v / n[(k, c)] / 1e6, 2) for c, v in sorted(agg[k].items())}, "launches", max(n[(k, c)] for c in agg[k]))
PY
rm -rf $OUT/p1 $OUT/p2
