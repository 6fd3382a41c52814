"""HBM traffic of the gemm_nt kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> "<command>"
Units/corrections per MI355X_MICROARCH.md (HBM section): counters are in KB; on gfx950 FETCH_SIZE
reports half of the bytes of wide (16 B/lane) coalesced reads -> x2; WRITE_SIZE is exact."""
import csv, glob, json, re, sys, collections


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if r["Counter_Name"] == counter and ("gemm_nt_kernel" in n or "gemm_nt8_kernel" in n or "gemm_nt_ares_kernel" in n or "bwd_fused_kernel" in n):
            key = re.search(r"(?:gemm_nt(?:_ares|8)?|bwd_fused)_kernel<[^>]*>", n).group(0)
            per[key].append(float(r["Counter_Value"]))
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
nf = sum(len(v) for v in fetch.values())
nw = sum(len(v) for v in write.values())
assert nf == nw and nf > 0, (nf, nw)
f_kb = sum(sum(v) for v in fetch.values()) / nf
w_kb = sum(sum(v) for v in write.values()) / nw
out = {
    "command": sys.argv[4],
    "kernel": "gemm_nt_kernel<*> + gemm_nt8_kernel<*> + gemm_nt_ares_kernel<*> + bwd_fused_kernel<*> (C entry points pcb_gemm_nt_bf16 / pcb_gemm_nt_red_bf16 / pcb_bwd_fused_bf16: the family bench.py times), all dispatches",
    "dispatches": nf,
    "FETCH_SIZE_KB_per_launch_raw": f_kb,
    "WRITE_SIZE_KB_per_launch": w_kb,
    "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) "
                  "coalesced reads -> x2; WRITE_SIZE exact for 16-B/lane stores; unit KB -> x1024",
    "traffic_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
    "per_variant_KB_per_launch": {
        k: {"FETCH_SIZE_raw": sum(fetch[k]) / len(fetch[k]), "WRITE_SIZE": sum(write[k]) / len(write[k]),
            "launches": len(fetch[k])} for k in sorted(fetch)},
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("dispatches", "traffic_bytes_per_launch")}))
