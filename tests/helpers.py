"""Shared comparison helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def knn_tie_tolerant_mismatch(idx_a, idx_b, dist_of, rtol=0.0, atol=0.0):
    """Rows where two kNN index lists differ by more than a permutation of equal
    (or, with tolerances, nearly equal) distances.

    dist_of(b, i, js) -> distances of candidates js for query (b, i).
    A row passes if the sorted distance lists agree (within tolerance): then the two lists
    pick the same neighbours up to ties.  Returns the number of failing rows.
    """
    B, N, k = idx_a.shape
    bad = 0
    for b in range(B):
        for i in range(N):
            a, c = idx_a[b, i], idx_b[b, i]
            if np.array_equal(a, c):
                continue
            da = np.sort(dist_of(b, i, a))
            dc = np.sort(dist_of(b, i, c))
            if not np.allclose(da, dc, rtol=rtol, atol=atol):
                bad += 1
    return bad


def free_port():
    """A TCP port that is free right now on 127.0.0.1 (bind to port 0): two suites on one box must not meet on a fixed
    rendezvous port."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p
