#!/usr/bin/env python3
"""Throughput of the secondary operators on synthetic batches (device-resident inputs):
ols_iter / pearson_corr / chisq_test from counts (BASELINE config 2: 100 pools x 1M loci) and
gp::ols (X X^T + coefficient pass).  Prints one JSON line per operator."""
import json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poolgen_amd import Engine, Filter, synth

def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    eng = Engine(0)
    counts = synth.sync_counts(L, n, "cuda")
    G = synth.genotype_matrix(L, n, "cuda")
    Y = synth.phenotypes(G, n, k=1)
    ps = np.full(n, 20.0); f = Filter()
    eng.profile(True)
    for name, fn, kid in (("ols_iter", lambda: eng.ols_iterate(counts, ps, f, Y, raw=True), "ols_iter"),
                          ("pearson_corr", lambda: eng.correlation(counts, ps, f, Y, raw=True), "pearson"),
                          ("chisq_test", lambda: eng.chisq(counts, ps, f, raw=True), "chisq")):
        fn(); fn(); fn()
        eng.profile_reset()
        dt = timeit(fn, reps=20)
        ms, cnt = eng.profile_get(kid)
        kms = ms / max(cnt, 1)
        print(json.dumps({"op": name, "pools": n, "loci": L, "wall_ms": dt * 1e3, "kernel_ms": kms,
                          "loci_per_s": L / dt, "gbs_algorithmic": 24.0 * n * L / (kms * 1e-3) / 1e9,
                          "frac_of_hbm_peak": 24.0 * n * L / (kms * 1e-3) / 8e12}))
    # the loader (counts -> G): plan + emit, all loci survive with two alleles each on this synthetic batch
    t = timeit(lambda: eng.load_frequencies(counts, ps, f), reps=3)
    Gl, _, _ = eng.load_frequencies(counts, ps, f)
    moved = 2 * 24.0 * n * L + 8.0 * Gl.shape[0] * Gl.shape[1]   # counts are read by the plan and by the emit pass
    print(json.dumps({"op": "load_frequencies", "pools": n, "loci": L, "columns": int(Gl.shape[0]), "wall_ms": t * 1e3,
                      "loci_per_s": L / t, "gbs_moved": moved / t / 1e9}))
    del Gl
    if G.shape[1] != n:
        return   # gp_ols below assumes ld == n
    idx = np.arange(n)
    eng.profile_reset()
    dt = timeit(lambda: eng.gp_ols(G, Y, idx), reps=3)
    x_ms, x_n = eng.profile_get("gp_xxt"); b_ms, b_n = eng.profile_get("gp_beta")
    print(json.dumps({"op": "gp_ols", "pools": n, "loci": L, "wall_ms": dt * 1e3, "xxt_ms": x_ms / max(x_n, 1),
                      "beta_ms": b_ms / max(b_n, 1), "loci_per_s": L / dt}))

if __name__ == "__main__":
    main()
