import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from poolgen_amd import Engine, Filter, synth
n, L = 100, 1_000_000
eng = Engine(0)
counts = synth.sync_counts(L, n, "cuda")
G = synth.genotype_matrix(1 << 18, n, "cuda"); Y = synth.phenotypes(G, n, k=1); del G
ps = np.full(n, 20.0); f = Filter()
for frac in (0.0, 0.05, 0.5, 1.0):
    c = counts.clone()
    nsel = int(L * frac)
    if nsel:
        idx = torch.randperm(L, device="cuda")[:nsel]
        c[idx, 7, 2] = 1          # one read of a third allele in pool 7: dropped by the MAF filter WITH reads -> second pass
    eng.profile(True)
    for name, fn, kid in (("ols_iter", lambda: eng.ols_iterate(c, ps, f, Y, raw=True), "ols_iter"), ("chisq", lambda: eng.chisq(c, ps, f, raw=True), "chisq")):
        fn(); fn(); eng.profile_reset()
        for _ in range(5): fn()
        ms, cnt = eng.profile_get(kid)
        print(f"deferred fraction {frac}: {name} {ms/cnt:.3f} ms")
