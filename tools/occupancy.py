"""Resident workgroups per CU that LDS and registers allow each kernel of a `rocprofv3 --kernel-trace --output-format csv` run
(160 KB of LDS, 512 VGPRs per SIMD lane, 4 SIMDs per CU):  python tools/occupancy.py <dir with *_kernel_trace.csv> [substring ...]"""
import csv
import glob
import sys

rows = list(csv.DictReader(open(sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0])))
want = sys.argv[2:]
seen = {}
for r in rows:
    n = r["Kernel_Name"]
    if want and not any(w in n for w in want):
        continue
    k = (n, r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])
    d = seen.setdefault(k, [0, 0.0])
    d[0] += 1
    d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (n, wg, lds, v, a), (c, us) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
    wg, lds, regs = int(wg), int(lds), int(v) + int(a)
    waves = max(wg // 64, 1)
    by_lds = 160 * 1024 // lds if lds else 99
    by_reg = (512 // max(regs, 1)) * 4 // waves
    short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{us:10.1f} us {c:5d}x  wg {wg:4d} lds {lds:6d} regs {regs:3d}  WG/CU: LDS {by_lds:2d} regs {by_reg:2d} -> waves/SIMD {min(by_lds, by_reg) * waves / 4:4.1f}  {short}")
