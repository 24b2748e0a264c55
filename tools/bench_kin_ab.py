#!/usr/bin/env python3
"""A/B of the 13-tile kinship kernel variants in one process: tools/bench_kin_ab.py [pools] [loci] [reps]"""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poolgen_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eng = Engine(0)
G = synth.genotype_matrix(p, n, "cuda")
Y = synth.phenotypes(G[:100000], n, k=1)
ref = None
for rnd in range(2):
    for fused in (True, False):
        for small in (0, 1, 2, 3):
            os.environ["POOLGEN_KIN_SMALL"] = str(small)
            eng.set_phenotypes(Y if fused else None)
            S = eng.kinship_partial(G, n); torch.cuda.synchronize()
            if ref is None: ref = S.clone()
            err = float(((S - ref).abs() / ref.abs()).max())
            eng.profile(True); eng.profile_reset()
            for _ in range(reps): eng.kinship_partial(G, n)
            ms, cnt = eng.profile_get("kinship"); eng.profile(False)
            t = ms / cnt
            tiles = (n + 15) // 16
            print(f"round {rnd} fused={int(fused)} small={small}: {t:7.3f} ms  useful {n * (n + 1) * p / t / 1e9:6.1f} TFLOP/s = {n * (n + 1) * p / t / 1e9 / 78.6:.3f} of peak   max rel diff to the first variant {err:.1e}")
