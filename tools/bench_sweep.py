#!/usr/bin/env python3
"""A/B timing of k_ols_sweep variants in ONE process (same clocks, same data): tools/bench_sweep.py [pools] [loci] [reps]
Variants are selected by environment variables the library reads at launch time."""
import os, sys, json
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poolgen_amd import Engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
variants = [v for v in os.environ.get("SWEEP_VARIANTS", "mf,v2,v1").split(",")]
eng = Engine(0)
G = synth.genotype_matrix(p, n, "cuda")
Y = synth.phenotypes(G[:100000], n, k=1)
out = torch.empty((3, p, 1), dtype=torch.float64, device="cuda")
rng = np.random.default_rng(1)
res = {}
for m in [int(x) for x in os.environ.get('SWEEP_MS', '0,8').split(',')]:
    C = None if m == 0 else np.linalg.qr(rng.normal(size=(n, m)))[0]
    eng.covariates_set(n, C, Y)
    ref = None
    for rnd in range(2):
        for v in variants:
            for key in ("POOLGEN_SWEEP_V1", "POOLGEN_SWEEP_V2", "POOLGEN_SWEEP_GRID_MULT", "POOLGEN_SWEEP_NOPF", "POOLGEN_SWEEP_MODE", "POOLGEN_SWEEP_EXP", "POOLGEN_SWEEP_U", "POOLGEN_SWEEP_R"):
                os.environ.pop(key, None)
            if v == "v1":
                os.environ["POOLGEN_SWEEP_V1"] = "1"
            elif v == "v2":
                os.environ["POOLGEN_SWEEP_V2"] = "1"
            elif v.startswith("v2exp"):
                os.environ["POOLGEN_SWEEP_V2"] = "1"
                os.environ["POOLGEN_SWEEP_EXP"] = v[5:]
            elif v.startswith("v2mode"):
                os.environ["POOLGEN_SWEEP_V2"] = "1"
                os.environ["POOLGEN_SWEEP_MODE"] = v[6:]
            elif v == "v2nopf":
                os.environ["POOLGEN_SWEEP_V2"] = "1"
                os.environ["POOLGEN_SWEEP_NOPF"] = "1"
            elif v.startswith("v2g"):
                os.environ["POOLGEN_SWEEP_V2"] = "1"
                os.environ["POOLGEN_SWEEP_GRID_MULT"] = v[3:]
            elif v.startswith("mfu"):      # "mf" = the product's default (the matrix-core sweep); mfu<U>: chunks per load group
                os.environ["POOLGEN_SWEEP_U"] = v[3:]
            elif v.startswith("mfexp"):    # timing experiments (wrong results)
                os.environ["POOLGEN_SWEEP_EXP"] = v[5:]
            elif v.startswith("mfr"):      # mfr<ring depth>
                os.environ["POOLGEN_SWEEP_R"] = v[3:]
            elif v.startswith("mfg"):      # mfg<blocks per CU>
                os.environ["POOLGEN_SWEEP_GRID_MULT"] = v[3:]
            eng.ols_sweep(G, 1, n, out); torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            elif "mode" not in v and "exp" not in v:
                assert torch.equal(torch.isnan(out[0]), torch.isnan(ref[0])), (v, "NaN pattern")   # (flat columns: NaN on both sides)
                d = float(torch.nan_to_num(out[0] - ref[0]).abs().max()); dp = float(torch.nan_to_num(out[2] - ref[2]).abs().max())
                assert d < 1e-9 and dp < 1e-9, (v, d, dp)
            eng.profile(True); eng.profile_reset()
            for _ in range(reps):
                eng.ols_sweep(G, 1, n, out)
            ms, cnt = eng.profile_get("sweep")
            eng.profile(False)
            res.setdefault(f"m{m}_{v}", []).append(ms / cnt)
for k, v in res.items():
    ms = min(v)
    print(f"{k:10s} {ms:7.3f} ms  {(8 * n + 24) * p / ms / 1e6:8.1f} GB/s  {(8 * n + 24) * p / ms / 1e6 / 8000:.3f} of 8 TB/s   rounds {['%.3f' % x for x in v]}")
