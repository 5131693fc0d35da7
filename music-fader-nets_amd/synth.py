"""Synthetic batches with the shapes / dtypes / value ranges of the reference's data loaders
(mirrors ptb_v2.py:261,264,319 (EOS + zero padding), :352-356 (chroma), :421-422 (densities),
:472-473 (binary arousal label)).  There is no dataset in this environment; benchmarks say ``"data": "synthetic"``.
"""
import numpy as np


def synth_batch(rng, B, T, Tr):
    d = np.zeros((B, T), np.int64)
    for b in range(B):
        L = int(rng.randint(T // 2, T + 1))
        d[b, :L - 1] = rng.randint(2, 342, size=L - 1)
        d[b, L - 1] = 1
    r = rng.choice(3, size=(B, Tr), p=[.3, .4, .3]).astype(np.int64)
    r[:, 0] = 1
    n = rng.randint(0, 14, size=(B, Tr)).astype(np.int64)
    c = np.zeros((B, 24), np.float32)
    for b in range(B):
        k = int(rng.randint(1, 4))
        pos = rng.choice(24, size=k, replace=False)
        c[b, pos] = rng.uniform(0.1, 1.0, size=k).astype(np.float32)
    r_density = np.array([(row == 1).sum() / Tr for row in r], np.float64)
    n_density = n.mean(axis=1).astype(np.float64)
    a = rng.randint(0, 2, size=(B,)).astype(np.int64)
    return dict(d=d, r=r, n=n, c=c, r_density=r_density, n_density=n_density, a=a)
