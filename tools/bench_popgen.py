"""Throughput of fst / theta_pi on a resident synthetic matrix: python tools/bench_popgen.py [pools] [loci]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from poolgen_amd import Engine, Filter, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
eng = Engine(0)
counts = synth.sync_counts(L, n, "cuda", seed=3)
ps = np.full(n, 20.0)
G, col_locus, col_allele, cov = eng.load_frequencies(counts, ps, Filter(), coverages=True)
del counts
cl = col_locus.cpu().numpy()
starts = np.flatnonzero(np.r_[True, cl[1:] != cl[:-1]]).tolist() + [len(cl)]
nl = len(starts) - 1
pos = np.arange(nl, dtype=np.uint64) * 37 + 100
chrom = (np.arange(nl) // (nl // 4 + 1)).astype(np.int32)
wh, wt = eng.sliding_windows(chrom, pos, 37 * 400, 37 * 200, 10)      # ~400 loci per window, half-overlapping
for name, fn in (("theta_pi", lambda: eng.theta_pi(G, cov, starts, wh, wt, n=n)), ("fst", lambda: eng.fst(G, cov, starts, wh, wt, n=n))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pairs = n * (n + 1) // 2
    print(json.dumps({"op": name, "pools": n, "loci": nl, "columns": int(G.shape[0]), "windows": int(len(wh)), "wall_s": dt,
                      "loci_per_s": nl / dt, "pair_locus_evals_per_s": (3 * nl * pairs / dt) if name == "fst" else None}))
