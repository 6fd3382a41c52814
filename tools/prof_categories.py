"""Kernel time per step by family from a `rocprofv3 --kernel-trace --stats` run: prof_categories.py <dir> <steps>.
<steps> = 0: the number of executed steps is taken from the once-per-step loss kernel (ce_fwd_kernel)."""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv") + glob.glob(sys.argv[1] + "/*_kernel_stats.csv"))[0]
steps = int(sys.argv[2])
FAMILIES = (("bwd_fused", "bwd_fused (dx + dW + sums, one pass)"), ("gemm_nt", "gemm_nt"), ("gemm_tn", "gemm_tn"), ("fps_", "fps (side stream)"), ("knn_", "knn"), ("reduce_slabs", "reduce_slabs"),
            ("bn_finalize", "bn_finalize"), ("bn_bwd_finalize", "bn_bwd_finalize"), ("prep_", "prep_weights / prep_linear_bias"),
            ("bn_act_bwd_reduce", "bn_act_bwd_reduce"), ("bn_max_bwd", "bn_max_bwd"), ("bn_act_max", "bn_act_max"),
            ("bn_act_kernel", "bn_act"), ("scatter_dy", "scatter_dy"), ("gather_add", "gather_add"), ("sum_slabs", "sum_slabs"),
            ("ball_query", "ball_query"), ("three_nn", "three_nn"), ("interp", "interpolate*"), ("csr_", "interpolate*"), ("gate", "gate*"),
            ("colstats", "colstats"), ("group_rows", "group_rows*"), ("gather_rows", "gather_rows"), ("nbr_", "nbr_mlp"),
            ("rows_linear", "rows_linear"), ("structure_features", "structure_features"), ("edge_features", "edge_features"),
            ("at::native", "ATen"), ("rocprim", "ATen"), ("Cijk", "hipBLASLt"), ("copyBuffer", "memcpy"), ("fillBuffer", "memset"), ("copy_list", "copy_table / copy_list"), ("copy_table", "copy_table / copy_list"),
            ("ce_", "cross entropy"), ("cew_", "cross entropy"), ("bridge_", "bridge loss weights"), ("rows_bn", "narrow BatchNorm rows"), ("scene_sum", "narrow BatchNorm rows"),
            ("segment_sum", "segment sums (reproducible mode)"), ("dropout", "dropout"), ("dy_repeat", "dy_repeat_sums / dy_rows"), ("dy_rows", "dy_repeat_sums / dy_rows"),
            ("zero2", "zero / pad"), ("zero_kernel", "zero / pad"), ("pad_rows", "zero / pad"), ("scene_", "scene pool / concat"), ("repeat_concat", "repeat_concat"), ("attention", "attention"), ("add_layernorm", "token rows (add+LayerNorm, GEGLU)"), ("geglu", "token rows (add+LayerNorm, GEGLU)"))
tot = {}
rows = list(csv.DictReader(open(f)))
if steps == 0:
    steps = max([int(r["Calls"]) for r in rows if "ce_fwd_kernel" in r["Name"] or "cew_fwd_kernel" in r["Name"]] or [1])
for r in rows:
    fam = next((v for k, v in FAMILIES if k in r["Name"]), "other: " + r["Name"][:50])
    d = tot.setdefault(fam, [0, 0])
    d[0] += int(r["TotalDurationNs"]); d[1] += int(r["Calls"])
print(f"total {sum(v[0] for v in tot.values()) / steps / 1e6:.3f} ms/step in {sum(v[1] for v in tot.values()) / steps:.0f} launches")
for fam, (t, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{t / steps / 1e6:7.3f} ms {n / steps:6.1f} launches  {fam}")
