"""Instruction mix of the main loop of chosen kernels in a hipcc -S --cuda-device-only listing:
    python tools/isa_mix.py file.s <substring of mangled name> ...
Prints per kernel: instructions in the whole body and inside its hottest loop (the innermost loop label range with
the most MFMAs), split into VALU / packed VALU / MFMA / LDS / global / scalar / waits."""
import collections
import re
import sys


def kernels(txt):
    for m in re.finditer(r"\n(_Z[^\n:]+):[^\n]*\n(.*?)\n\s+s_endpgm", txt, flags=re.S):
        yield m.group(1), m.group(2)


def mix(lines):
    ops = collections.Counter()
    for l in lines:
        l = l.strip()
        m = re.match(r"([a-z_0-9]+)\s", l + " ")
        if m and not l.startswith((".", ";")) and not l.endswith(":"):
            ops[m.group(1)] += 1
    g = lambda pred: sum(c for o, c in ops.items() if pred(o))
    return {"total": sum(ops.values()), "valu": g(lambda o: o.startswith("v_") and not o.startswith("v_mfma")),
            "pk": g(lambda o: o.startswith("v_pk")), "mfma": g(lambda o: o.startswith("v_mfma")),
            "ds": g(lambda o: o.startswith("ds_")), "vmem": g(lambda o: o.startswith(("global_", "buffer_", "scratch_"))),
            "salu": g(lambda o: o.startswith("s_") and o not in ("s_waitcnt", "s_barrier", "s_nop")),
            "wait": ops["s_waitcnt"], "barrier": ops["s_barrier"], "nop": ops["s_nop"]}, ops


def main():
    txt = open(sys.argv[1]).read()
    for name, body in kernels(txt):
        if not any(k in name for k in sys.argv[2:]):
            continue
        lines = body.splitlines()
        m, ops = mix(lines)
        print(name[:90])
        print("  body:", m)
        # loops: backward branches to a label
        labels = {}
        for i, l in enumerate(lines):
            lm = re.match(r"\s*(\.LBB\d+_\d+):", l)
            if lm:
                labels[lm.group(1)] = i
        best = None
        for i, l in enumerate(lines):
            b = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s*s_branch\s+(\.LBB\d+_\d+)", l)
            if b and b.group(1) in labels and labels[b.group(1)] < i:
                seg = lines[labels[b.group(1)]:i + 1]
                mm, oo = mix(seg)
                if mm["mfma"] and (best is None or mm["total"] < best[0]["total"]):
                    best = (mm, oo)
        if best:
            print("  innermost MFMA loop:", best[0])
            print("   ", [(o, c) for o, c in best[1].most_common(28) if o.startswith("v_")])


main()
