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


def _spread_errors(src: torch.Tensor, rate: float, g: torch.Generator):
    """Sequencing errors of the reads in `src` (float64 counts): each read is misread with probability `rate`, and a misread
    read lands on one of the FIVE other sync columns with equal probability (the other allele, the two other bases, N, D).
    Returns (reads that stay, [reads that move to each of the 5 other columns])."""
    moved = torch.binomial(src, torch.full_like(src, rate), generator=g)
    parts, rest = [], moved
    for left in (5, 4, 3, 2):
        part = torch.binomial(rest, torch.full_like(rest, 1.0 / left), generator=g)
        parts.append(part)
        rest = rest - part
    parts.append(rest)
    return src - moved, parts


def sync_counts(p: int, n: int, device, seed: int = SEED, start: int = 0, error_rate: float = 0.0):
    """Sync-style counts (p x n x 6 int32, columns A,T,C,G,N,D): A = ref, T = alt, rest 0.

    error_rate > 0 makes the counts look like real pool-seq data: every read is misread with that probability and lands on one
    of the other five columns (so a biallelic A/T locus carries stray C, G, N and D reads -- about n * depth * error_rate * 3/5 of
    them per locus outside A, T and N).  The error draws come after the clean draws of a chunk, so the clean part is the same
    matrix as with error_rate = 0."""
    out = torch.zeros((p, n, 6), dtype=torch.int32, device=device)
    lo = start
    while lo < start + p:
        c = lo // CHUNK
        c_lo, c_hi = c * CHUNK, (c + 1) * CHUNK
        hi = min(start + p, c_hi)
        alt, depth = _chunk(seed, c, CHUNK, n, device, True)
        ref = depth - alt
        if error_rate > 0.0:
            g = torch.Generator(device=device)
            g.manual_seed(seed * 1_000_003 + c + 500_000_009)
            ref_stay, ref_to = _spread_errors(ref, error_rate, g)   # A -> T, C, G, N, D
            alt_stay, alt_to = _spread_errors(alt, error_rate, g)   # T -> A, C, G, N, D
            cols = [ref_stay + alt_to[0], alt_stay + ref_to[0]] + [ref_to[1 + j] + alt_to[1 + j] for j in range(4)]
        else:
            cols = [ref, alt]
        for j, col in enumerate(cols):
            out[lo - start:hi - start, :, j] = col[lo - c_lo:hi - c_lo].to(torch.int32)
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
