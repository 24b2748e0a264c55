#!/usr/bin/env python3
"""Count operators on realistic (error-bearing) counts: kernel time per launch, fraction of the HBM peak, listed fraction.
usage: bench_ops_realistic.py [pools] [loci] [error_rate maf]   -> JSON lines (profiles/r04_ops_realistic.jsonl)"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from poolgen_amd import Engine, Filter, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
eng = Engine(0)
G = synth.genotype_matrix(1 << 18, n, "cuda"); Y = synth.phenotypes(G, n, k=1); del G
ps = np.full(n, 20.0)
eng.profile(True)
cases = ((0.0, 0.001), (0.001, 0.001), (0.005, 0.01), (0.01, 0.01), (0.005, 0.001))
if len(sys.argv) > 4:
    cases = ((float(sys.argv[3]), float(sys.argv[4])),)
for err, maf in cases:
    c = synth.sync_counts(L, n, "cuda", error_rate=err)
    f = Filter(min_allele_frequency=maf)
    for name, fn, kid in (("ols_iter", lambda: eng.ols_iterate(c, ps, f, Y, raw=True), "ols_iter"),
                          ("pearson_corr", lambda: eng.correlation(c, ps, f, Y, raw=True), "pearson"),
                          ("chisq_test", lambda: eng.chisq(c, ps, f, raw=True), "chisq")):
        fn(); fn(); eng.profile_reset()
        for _ in range(10): fn()
        ms, cnt = eng.profile_get(kid)
        ms /= cnt
        by = 24.0 * n * L
        print(json.dumps({"pools": n, "loci": L, "error_rate": err, "min_allele_frequency": maf, "op": name, "kernel_ms": ms,
                          "frac_of_8TBs": by / (ms * 1e-3) / 8e12, "listed_fraction": eng.last_listed_fraction()}), flush=True)
    del c
