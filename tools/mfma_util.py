"""MFMA utilisation of the GEMM kernels from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass.

usage: mfma_util.py <pmc_dir> <out.json> "<command>"
utilisation = busy cycles / (kernel duration x shader clock x 1024 SIMDs); the counter adds up the
matrix-pipe busy cycles of all 256 CUs x 4 SIMDs (64 cycles per 32x32x2 f32 MFMA, 32 per 32x32x16 bf16)."""
import csv, glob, json, re, sys, collections
CLOCK_GHZ, SIMDS = 2.4, 1024
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
        continue
    m = re.search(r"(gemm_nt_ares_kernel|gemm_nt8_kernel|gemm_nt_kernel|gemm_tn_kernel|knn_mfma_kernel)<[^>]*>", r["Kernel_Name"])
    if not m:
        continue
    a = agg[m.group(0)]
    a[0] += float(r["Counter_Value"])
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[2] += 1
out = {"command": sys.argv[3], "clock_GHz_assumed": CLOCK_GHZ, "kernels": {}}
tot_b = tot_t = 0.0
for k, (busy, ns, n) in sorted(agg.items()):
    util = busy / (ns * CLOCK_GHZ * SIMDS)
    out["kernels"][k] = {"launches": n, "avg_us": ns / n / 1e3, "mfma_busy_cycles_per_launch": busy / n, "mfma_util": util}
    tot_b += busy; tot_t += ns
out["all_gemm_mfma_util"] = tot_b / (tot_t * CLOCK_GHZ * SIMDS)
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:34s} {v['launches']:4d} x {v['avg_us']:7.1f} us  MFMA util {v['mfma_util']*100:5.1f} %")
print("all", round(out["all_gemm_mfma_util"] * 100, 1), "%")
