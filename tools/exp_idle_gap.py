#!/usr/bin/env python3
"""usage (GPU box): tools/exp_idle_gap.py -- does the kinship pass run slower after the GPU sat idle for the host's eigen step?
200 pools x 10 M loci, the two-pass route with a host pause of 0 / 0.5 / 1.0 / 1.5 / 3 ms between the kinship pass and the sweep
(the forced-m = 8 step pauses 1.4 ms there for the n x n solve).  Prints the HIP-event means of both kernels per pause."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from poolgen_amd import Engine, synth
n, p = 200, 10_000_000
eng = Engine()
G = synth.genotype_matrix(p, n, "cuda")
Y = synth.phenotypes(G[:100000], n, k=1)
eng.set_phenotypes(None)
eng.profile(True)
for pause_ms in (0.0, 0.5, 1.0, 1.5, 3.0, 0.0):
    for rep in range(8):
        if rep == 2: eng.profile_reset()
        S = eng.kinship_partial(G, n)
        torch.cuda.synchronize()
        m, K, _ = eng.kinship_set(S, p, Y, 0.75, 0, want_K=False)
        if pause_ms: 
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < pause_ms: pass
        beta, var, pv = eng.ols_sweep(G, 1, n)
        torch.cuda.synchronize()
    print("pause %.1f ms: kinship %s  sweep %s" % (pause_ms, eng.profile_get("kinship"), eng.profile_get("sweep")), flush=True)
