#!/usr/bin/env python3
"""Does the sweep's rate depend on WHERE the genotype matrix lies?  One process, several copies of the same 200 x 10 M matrix:
separate allocations, and views into one allocation at different row offsets.  Prints the sweep kernel time for each."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from poolgen_amd import Engine, synth
n, p = 200, 10_000_000
eng = Engine(0); eng.profile(True)
G0 = synth.genotype_matrix(p, n, "cuda")
Y = synth.phenotypes(G0[:100000], n, k=1)
eng.covariates_set(n, np.ones((n, 0)), Y)        # intercept-only design, handed in: the sweep alone
def sweep_ms(G, reps=5):
    out = eng.ols_sweep(G, 1, n=n); torch.cuda.synchronize()
    eng.profile_reset()
    for _ in range(reps): eng.ols_sweep(G, 1, n=n, out=out)
    ms, cnt = eng.profile_get("sweep")
    return ms / cnt
print("first allocation            ptr %#x  %.3f ms" % (G0.data_ptr(), sweep_ms(G0)))
keep = [G0]
for i in range(4):
    G = torch.empty_like(G0); G.copy_(G0); keep.append(G)
    print("separate allocation %d       ptr %#x  %.3f ms" % (i + 1, G.data_ptr(), sweep_ms(G)))
del keep[1:]; torch.cuda.empty_cache()
big = torch.empty((p + 4096, n), dtype=torch.float64, device="cuda")
for off in (0, 1, 2, 16, 64, 640, 1280, 4095):
    V = big[off:off + p]; V.copy_(G0)
    print("view at row offset %-5d    ptr %#x  %.3f ms" % (off, V.data_ptr(), sweep_ms(V)))
print("first allocation again      ptr %#x  %.3f ms" % (G0.data_ptr(), sweep_ms(G0)))
