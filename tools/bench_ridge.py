#!/usr/bin/env python3
"""Wall time of the ridge lambda path with k-fold CV (BASELINE config 4 shape: 500 pools; loci and repetitions scaled by argv).
usage: bench_ridge.py [pools] [loci] [reps] [folds]"""
import json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poolgen_amd import Engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
p = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
folds = int(sys.argv[4]) if len(sys.argv) > 4 else 10
eng = Engine(0)
G = synth.genotype_matrix(p, n, "cuda")
Y = synth.phenotypes(G[:100000], n, k=1)
rows = np.arange(n)
fold_of = np.stack([(rows + r) % folds for r in range(reps)]).astype(np.int32)   # fold[i] = i mod 10
eng.profile(True)
for it in range(2):
    eng.profile_reset(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    beta, lam, perf = eng.gp_ridge(G, Y, rows, fold_of, folds, n=n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
x_ms, x_n = eng.profile_get("gp_xxt"); b_ms, b_n = eng.profile_get("gp_beta"); q_ms, q_n = eng.profile_get("gp_predict")
print(json.dumps({"op": "gp_ridge", "pools": n, "loci": p, "reps": reps, "folds": folds, "wall_s": dt, "lambda": lam.tolist(),
                  "xxt_ms_total": x_ms, "xxt_launches": x_n, "beta_ms_total": b_ms, "beta_launches": b_n, "predict_ms_total": q_ms, "predict_launches": q_n,
                  "predict_frac_of_8TBs": (8.0 * n * p * q_n / (q_ms * 1e-3) / 8e12) if q_ms else None,
                  "passes_over_G_equivalent_GBps": (2 * reps * folds + 2) * 8.0 * n * p / dt / 1e9}))
