#!/usr/bin/env python3
"""Does a light memory-bound kernel run BESIDE the kinship kernel (one 1024-thread workgroup per CU, 110 registers per lane,
53 KB of LDS) or only after it?  The non-fused kinship pass on one stream, a torch row reduction of the same matrix on another."""
import sys, time
sys.path.insert(0, ".")
import torch
from poolgen_amd import Engine, synth
n, p = 200, 10_000_000
eng = Engine(0)
G = synth.genotype_matrix(p, n, "cuda")
eng.set_phenotypes(None)
s2 = torch.cuda.Stream()
def kin(): return eng.kinship_partial(G, n)
def side(): return (G * G).sum(dim=1) if False else G.sum(dim=1)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("kinship alone %.3f ms" % t(kin))
print("row sums alone %.3f ms" % t(side))
def both():
    kin()
    with torch.cuda.stream(s2):
        side()
def both_rev():
    with torch.cuda.stream(s2):
        side()
    kin()
print("kinship then row sums on a second stream %.3f ms" % t(both))
print("row sums first on a second stream, then kinship %.3f ms" % t(both_rev))
