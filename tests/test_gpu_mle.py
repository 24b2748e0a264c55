"""mle_iter_with_kinship (gwas::mle_with_covariate, gwas/mle.rs:307-463) -- PARITY UNPINNED: the reference has no test of this path
(`fn test_mle() {}`, mle.rs:470) and its numbers are wherever argmin 0.8's Nelder-Mead simplex stands after <= 1000 iterations, a
crate whose source is not in the reference tree.  What CAN be checked, and is:
  * the GPU's simplex (on sufficient statistics) and the oracle's literal restatement of the same published solver around the
    reference's cost function agree at the solver's own resolution;
  * both sit at the analytic optimum of that cost: the OLS coefficient, and sigma^2 = 2 RSS / n (the cost has 1 / sigma^2, not
    1 / (2 sigma^2), mle.rs:27) -- so v_b = 2 (n - P) / n times the OLS variance;
  * the closing arithmetic as written (t = b / v_b, mle.rs:175) and the CLI's file (name, header, label shift)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CLI = ROOT / "poolgen_amd" / "csrc" / "poolgen"
GOLD = Path(__file__).parent / "golden"


def make(p, n, seed, k):
    from poolgen_amd import synth
    G = synth.genotype_matrix(p, n, "cuda", seed=seed)
    Y = synth.phenotypes(G, n, k=2, seed=seed)[:, :k]
    return G, Y


@pytest.mark.parametrize("n,p,k,force_m", [(60, 600, 1, 0), (100, 400, 2, 0), (40, 300, 1, 1), (200, 300, 1, 2)])
def test_mle_against_the_oracle_and_the_analytic_optimum(engine, oracle, n, p, k, force_m, capsys):
    G, Y = make(p, n, 101, k)
    Gh = G.cpu().numpy()
    m, K, beta, var, pv = engine.mle_with_covariate(G, Y, 0.75, force_m=force_m, n=n)
    beta, var, pv = beta.cpu().numpy(), var.cpu().numpy(), pv.cpu().numpy()
    assert m == force_m
    # the product's covariates are its own leading eigenvectors; hand the oracle eigenvectors of the same K
    w, Vv = np.linalg.eigh(K)
    C = Vv[:, ::-1][:, :m].copy() if m else None
    ref = oracle.mle_with_covariate(Gh, Y, covariate=C, force_m=m, n=n, threads=8) if m else oracle.mle_with_covariate(Gh, Y, force_m=0, n=n, threads=8)
    ols = oracle.ols_with_covariate(Gh, Y, covariate=C, n=n) if m else oracle.ols_with_covariate(Gh, Y, force_m=0, n=n)
    scale = np.abs(ols["beta"]).max()
    d_go = np.abs(beta - ref["beta"]).max() / scale
    d_g = np.abs(beta - ols["beta"]).max() / scale
    d_o = np.abs(ref["beta"] - ols["beta"]).max() / scale
    P = m + 2
    vb_opt = ols["var"] * 2.0 * (n - P) / n
    rv_g = np.abs(var / vb_opt - 1.0).max()
    rv_o = np.abs(ref["var"] / vb_opt - 1.0).max()
    with capsys.disabled():
        print(f"\n[mle n={n} p={p} k={k} m={m}] beta / max|beta|: |GPU - oracle| {d_go:.1e}  |GPU - optimum| {d_g:.1e}  |oracle - optimum| {d_o:.1e}   "
              f"var: rel |GPU / optimum - 1| {rv_g:.1e}  |oracle / optimum - 1| {rv_o:.1e}")
    # the simplex stops within ~1e-7 .. 1e-5 of the optimum; ill-conditioned [1 | v1 ..] designs stop further out, and WHERE in
    # that neighbourhood 1000 iterations leave it hangs on the last bits of the sufficient statistics (n = 200, m = 2: 4.7e-3
    # with the vector-ALU summation order of g'W, 5.4e-3 with the matrix-core order; the oracle's own stands at 1.2e-3)
    tol = 2e-5 if m == 0 else 1e-2
    assert d_g <= tol and d_o <= tol and d_go <= 2 * tol
    assert rv_g <= 50 * tol and rv_o <= 50 * tol
    # p-values as written: t = b / v_b (variance, not standard error), df = n - 1
    import ctypes
    t = np.abs(beta / var)
    want = np.array([2.0 * (1.0 - oracle.lib.orc_students_t_cdf(float(x), float(n - 1))) for x in t.reshape(-1)]).reshape(t.shape)
    assert np.max(np.abs(pv - want)) <= 1e-10


@pytest.mark.parametrize("n,p,k,force_m", [(60, 300, 2, 0), (40, 200, 1, 1), (120, 200, 1, 2)])
def test_lds_simplex_is_the_register_simplex(engine, n, p, k, force_m, monkeypatch):
    """m = 3 .. 8 (5 .. 10 design columns) run a kernel whose simplex lives in LDS behind a rank -> vertex table.  Forced onto the
    small designs the register kernel also takes, it must return the SAME BITS: same solver, same order of every sum."""
    G, Y = make(p, n, 77, k)
    a = engine.mle_with_covariate(G, Y, 0.75, force_m=force_m, n=n)
    monkeypatch.setenv("POOLGEN_MLE_LDS", "1")
    b = engine.mle_with_covariate(G, Y, 0.75, force_m=force_m, n=n)
    for x, y in zip(a[2:], b[2:]):
        assert torch.equal(x, y) or (torch.isnan(x) == torch.isnan(y)).all() and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))


@pytest.mark.parametrize("n,p,force_m", [(100, 200, 3), (200, 150, 5), (150, 120, 8)])
def test_mle_with_many_covariates(engine, oracle, n, p, force_m, capsys):
    """The reference's design is n x (2 + n_eigenvecs) (mle.rs:376) for whatever the eigen rule yields; -x 0.99 on real data gives
    3 and more.  With 5 .. 10 coefficients plus sigma^2 the 1000-iteration cap (mle.rs:98) ends the simplex long before it has
    converged -- for the oracle's literal restatement exactly as for the GPU: measured on MI355X, the worst cell stands 0.12 / 2.3 /
    4.0 times the largest coefficient away from the optimum for m = 3 / 5 / 8 in the oracle and 0.14 / 3.2 / 4.1 on the GPU.  What
    the reference prints for such designs is where ITS simplex happens to stand; what can be held here is: a finite result for
    every cell, the closing arithmetic as written, and both restatements of the solver stopping equally far out (worst-cell
    distances within a factor of 3 of each other)."""
    G, Y = make(p, n, 303, 1)
    Gh = G.cpu().numpy()
    m, K, beta, var, pv = engine.mle_with_covariate(G, Y, 0.75, force_m=force_m, n=n)
    beta, var, pv = beta.cpu().numpy(), var.cpu().numpy(), pv.cpu().numpy()
    assert m == force_m and np.isfinite(beta).all() and np.isfinite(var).all() and (var > 0).all()
    w, Vv = np.linalg.eigh(K)
    C = Vv[:, ::-1][:, :m].copy()
    ref = oracle.mle_with_covariate(Gh, Y, covariate=C, force_m=m, n=n, threads=8)
    ols = oracle.ols_with_covariate(Gh, Y, covariate=C, n=n)
    scale = np.abs(ols["beta"]).max()
    d_g = np.abs(beta - ols["beta"]).max() / scale
    d_o = np.abs(ref["beta"] - ols["beta"]).max() / scale
    with capsys.disabled():
        print(f"\n[mle n={n} p={p} m={m}] beta / max|beta|: |GPU - optimum| {d_g:.1e}  |oracle - optimum| {d_o:.1e}")
    assert d_g <= 3 * max(d_o, 1e-6) and d_o <= 3 * max(d_g, 1e-6)
    t = np.abs(beta / var)
    want = np.array([2.0 * (1.0 - oracle.lib.orc_students_t_cdf(float(x), float(n - 1))) for x in t.reshape(-1)]).reshape(t.shape)
    assert np.max(np.abs(pv - want)) <= 1e-10


def test_cli_mle_iter_with_kinship(tmp_path):
    out = tmp_path / "mle.csv"
    r = subprocess.run([str(CLI), "mle_iter_with_kinship", "-f", str(GOLD / "test.sync"), "-p", str(GOLD / "test.csv"), "--phen-value-col", "2,3",
                        "--n-threads", "2", "-x", "0.5", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ols = tmp_path / "ols.csv"
    subprocess.run([str(CLI), "ols_iter_with_kinship", "-f", str(GOLD / "test.sync"), "-p", str(GOLD / "test.csv"), "--phen-value-col", "2,3",
                    "--n-threads", "2", "-x", "0.5", "-o", str(ols)], check=True, capture_output=True)
    a, b = out.read_text().splitlines(), ols.read_text().splitlines()
    assert a[0] == b[0] == "#chr,pos,alleles,phenotype,statistic,pvalue" and len(a) == len(b) > 10000
    assert a[1].startswith("intercept,0,intercept,Pheno_0,")       # the same label shift (mle.rs:447-452 = ols.rs:421-425)
    close = 0
    for x, y in zip(a[1:], b[1:]):
        fx, fy = x.split(","), y.split(",")
        assert fx[:4] == fy[:4]
        bx, by = float(fx[4]), float(fy[4])
        if np.isnan(bx) or np.isnan(by):
            continue
        close += abs(bx - by) <= 1e-3 * max(1.0, abs(by))
    assert close > 0.9 * (len(a) - 1)        # 5 pools: most simplices reach the OLS coefficient, flat likelihoods stop early
    name = subprocess.run([str(CLI), "mle_iter_with_kinship", "-f", str(GOLD / "test.sync"), "-p", str(GOLD / "test.csv"), "-x", "0.5"],
                          capture_output=True, text=True, cwd=tmp_path).stdout.strip().splitlines()[-1]
    assert "-mle_iterative_xxt_" in name and name.endswith(".csv")
    Path(name).unlink()
