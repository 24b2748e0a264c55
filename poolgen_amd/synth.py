"""Synthetic pool-seq data of BASELINE.md section 3 (generator is ours; the reference ships none).

Per locus: base frequency b ~ Beta(0.5, 0.5) clipped to [0.02, 0.98]; per pool depth
d ~ Poisson(60) + 10; alt count ~ Binomial(d, clip(b + N(0, 0.08), 0, 1)); G = alt / d (fp64).
Loci are generated in fixed-size chunks, each from its own generator seeded by
(seed, global chunk index), so any shard of the matrix is reproducible without the rest.
"""
from __future__ import annotations

import numpy as np
import torch

SEED = 20251003
CHUNK = 1 << 18  # loci per generator chunk


def _chunk(seed: int, c: int, rows: int, n: int, device, want_counts: bool):
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1_000_003 + c)
    # Beta(0.5, 0.5) = sin^2(pi/2 * U)  (arcsine law) -- exact and cheap
    u = torch.rand(rows, 1, generator=g, device=device, dtype=torch.float64)
    b = torch.sin(0.5 * torch.pi * u).square().clamp_(0.02, 0.98)
    pr = (b + 0.08 * torch.randn(rows, n, generator=g, device=device, dtype=torch.float64)).clamp_(0.0, 1.0)
    depth = torch.poisson(torch.full((rows, n), 60.0, device=device, dtype=torch.float64), generator=g) + 10.0
    alt = torch.binomial(depth, pr, generator=g)
    if want_counts:
        return alt, depth
    return alt / depth


def genotype_matrix(p: int, n: int, device, seed: int = SEED, start: int = 0, ld: int | None = None):
    """Locus-major allele-frequency matrix G[start:start+p] (p x ld fp64, columns >= n are zero)."""
    ld = n + (n & 1) if ld is None else ld
    G = torch.zeros((p, ld), dtype=torch.float64, device=device) if ld != n else \
        torch.empty((p, n), dtype=torch.float64, device=device)
    lo = start
    while lo < start + p:
        c = lo // CHUNK
        c_lo, c_hi = c * CHUNK, (c + 1) * CHUNK
        hi = min(start + p, c_hi)
        blk = _chunk(seed, c, CHUNK, n, device, False)
        G[lo - start:hi - start, :n] = blk[lo - c_lo:hi - c_lo]
        lo = hi
    return G


def sync_counts(p: int, n: int, device, seed: int = SEED, start: int = 0):
    """Sync-style counts (p x n x 6 int32, columns A,T,C,G,N,D): A = ref, T = alt, rest 0."""
    out = torch.zeros((p, n, 6), dtype=torch.int32, device=device)
    lo = start
    while lo < start + p:
        c = lo // CHUNK
        c_lo, c_hi = c * CHUNK, (c + 1) * CHUNK
        hi = min(start + p, c_hi)
        alt, depth = _chunk(seed, c, CHUNK, n, device, True)
        out[lo - start:hi - start, :, 1] = alt[lo - c_lo:hi - c_lo].to(torch.int32)
        out[lo - start:hi - start, :, 0] = (depth - alt)[lo - c_lo:hi - c_lo].to(torch.int32)
        lo = hi
    return out


def phenotypes(G_head: torch.Tensor, n: int, k: int = 1, seed: int = SEED, h2: float = 0.5):
    """k traits from 10 'causal' loci among the rows of G_head, heritability h2 (host ndarray n x k)."""
    rng = np.random.default_rng(seed)
    p = G_head.shape[0]
    idx = [(p * (2 * i + 1)) // 20 for i in range(10)]
    X = G_head[idx, :n].T.cpu().numpy()  # n x 10
    Y = np.empty((n, k))
    for j in range(k):
        beta = rng.normal(size=10)
        gval = X @ beta
        vg = gval.var()
        noise = rng.normal(size=n) * np.sqrt(vg * (1 - h2) / h2 if vg > 0 else 1.0)
        Y[:, j] = gval + noise
    return Y
