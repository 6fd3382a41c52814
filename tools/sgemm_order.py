"""Which accumulation orders does this container's CPU path use for the two reductions of the reference's
DGCNN.knn at D = 64 (models/DGCNN.py:60-65)?

    x = x.transpose(2, 1).contiguous()                   # [B, N, D]
    inner = -2 * torch.matmul(x, x.transpose(2, 1))      # MKL sgemm, K = D
    xx = torch.sum(x ** 2, dim=2, keepdim=True)          # ATen reduction over the contiguous axis
    pairwise_distance = xx + inner + xx.transpose(2, 1)

Candidate orders are evaluated in exact fp32 emulation (products of fp32 numbers are exact in fp64; one rounding
per operation) and compared BITWISE with torch's results.     python tools/sgemm_order.py
"""
import itertools

import numpy as np
import torch

D = 64


def f32(a):
    return a.astype(np.float32)


def fma32(acc, a, b):
    return f32(acc.astype(np.float64) + a.astype(np.float64) * b.astype(np.float64))


def mul32(a, b):
    return f32(a.astype(np.float64) * b.astype(np.float64))


def chain(A, Bt, order, use_fma=True):
    """sum_k A[:,k] * Bt[:,k]^T accumulated sequentially over k in `order`; A [n,D], Bt [m,D]"""
    acc = None
    for k in order:
        a, b = A[:, k:k + 1], Bt[:, k:k + 1].T
        if acc is None:
            acc = mul32(a, b)
        elif use_fma:
            acc = fma32(acc, a, b)
        else:
            acc = f32(acc + mul32(a, b))
    return acc


def combine(parts, tree):
    parts = list(parts)
    if tree:
        while len(parts) > 1:
            parts = [f32(parts[i] + parts[i + 1]) for i in range(0, len(parts), 2)]
        return parts[0]
    acc = parts[0]
    for p in parts[1:]:
        acc = f32(acc + p)
    return acc


def index_sets(ways, interleaved):
    return [list(range(w, D, ways)) if interleaved else list(range(w * D // ways, (w + 1) * D // ways)) for w in range(ways)]


def report(title, ref, cands, top=5):
    print(title)
    for name, c in sorted(cands.items(), key=lambda kv: -np.mean(kv[1].view(np.uint32) == ref.view(np.uint32)))[:top]:
        print(f"   {np.mean(c.view(np.uint32) == ref.view(np.uint32)):8.5f}  {name}")


def main():
    torch.manual_seed(0)
    print([l.strip() for l in torch.__config__.show().split("\n") if "Math Kernel" in l or "CPU capability" in l])
    for N in (64, 1000, 2048):
        x = torch.randn(2, D, N).transpose(2, 1).contiguous()     # [B, N, D] as in the reference
        X = x[0].numpy()
        ref = torch.matmul(x, x.transpose(2, 1))[0].numpy()
        cands = {"one fma chain, k = 0..63": chain(X, X, range(D)),
                 "one chain, k = 0..63, multiply then add": chain(X, X, range(D), use_fma=False),
                 "one fma chain, k = 63..0": chain(X, X, range(D - 1, -1, -1))}
        for ways, inter, tree in itertools.product((2, 4, 8, 16), (True, False), (False, True)):
            name = f"{ways} fma chains ({'interleaved' if inter else 'blocked'} k), {'tree' if tree else 'left-to-right'} sum"
            cands[name] = combine([chain(X, X, ks) for ks in index_sets(ways, inter)], tree)
        report(f"N={N}: inner products x x^T bit-identical to torch.matmul", ref, cands)

        ref_xx = torch.sum(x ** 2, dim=2)[0].numpy()
        sq = mul32(X, X)
        c2 = {"one chain, c = 0..63": combine([sq[:, c] for c in range(D)], False)}
        for ways, inter, tree in itertools.product((2, 4, 8, 16, 32), (True, False), (False, True)):
            parts = [combine([sq[:, c] for c in ks], False) for ks in index_sets(ways, inter)]
            c2[f"{ways} chains ({'interleaved' if inter else 'blocked'} c), {'tree' if tree else 'left-to-right'} sum"] = combine(parts, tree)
        # a vector of 8 / 16 lanes accumulating consecutive chunks, lanes then reduced horizontally
        for lanes, unroll in itertools.product((8, 16), (1, 2, 4)):
            accs = []
            for u in range(unroll):
                a = np.zeros((X.shape[0], lanes), np.float32)
                for c0 in range(u * lanes, D, lanes * unroll):
                    a = f32(a + sq[:, c0:c0 + lanes])
                accs.append(a)
            v = combine(accs, False)
            for tree in (False, True):
                if tree:
                    w = v
                    while w.shape[1] > 1:
                        h = w.shape[1] // 2
                        w = f32(w[:, :h] + w[:, h:])
                    r = w[:, 0]
                else:
                    r = combine([v[:, l] for l in range(lanes)], False)
                c2[f"{lanes}-lane vector x {unroll} accumulators over consecutive chunks, {'halving' if tree else 'left-to-right'} lane sum"] = r
        report(f"N={N}: |x|^2 bit-identical to torch.sum(x**2, dim=2)", ref_xx, c2)


if __name__ == "__main__":
    main()
