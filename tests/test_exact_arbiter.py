"""CPU tests of the binary128 arbiter (oracle/poolgen_exact.c) -- the reference point the fp64 covariate fits and the
gp::ols family are measured against on the GPU box (tests/test_gpu_exact.py).  Pinned here against mpmath at 50 digits
(an independent multiprecision implementation), on sizes mpmath finishes in seconds."""
import numpy as np
import pytest

mp = pytest.importorskip("mpmath")


def synth_G(p, n, seed):
    import torch
    from poolgen_amd import synth
    return synth.genotype_matrix(p, n, "cpu", seed=seed).numpy()[:, :n].copy()


def test_student_t_tail_against_mpmath(exact):
    mp.mp.dps = 50
    for df in (1, 2, 3, 4, 7, 59, 199, 200, 499):
        for t in (1e-9, 0.01, 0.3849, 1.0, 2.5, 7.0, 19.0, 40.0):
            h = mp.mpf(df) / (mp.mpf(df) + mp.mpf(t) ** 2)
            want = mp.betainc(mp.mpf(df) / 2, mp.mpf(1) / 2, 0, h, regularized=True)   # = P(|T| > t)
            got = exact.t_two_sided_p(t, df)
            assert abs(got - float(want)) <= 4e-16 * max(float(want), 1e-300) + 1e-30, (df, t, got, float(want))


def test_covariate_fit_against_mpmath(exact):
    mp.mp.dps = 50
    n, m = 12, 2
    rng = np.random.default_rng(5)
    G = rng.uniform(0.05, 0.95, size=(3, n))
    C = np.column_stack([np.full(n, 1 / np.sqrt(n)) * (1 + 1e-6 * rng.normal(size=n)), rng.normal(size=n)])  # nearly collinear
    Y = rng.normal(size=(n, 2))
    got = exact.ols_covariate(G, Y, C)
    for l in range(3):
        X = mp.matrix(n, m + 2)
        for i in range(n):
            X[i, 0] = 1
            for a in range(m):
                X[i, 1 + a] = mp.mpf(float(C[i, a]))
            X[i, m + 1] = mp.mpf(float(G[l, i]))
        inv = (X.T * X) ** -1
        for j in range(2):
            y = mp.matrix([mp.mpf(float(v)) for v in Y[:, j]])
            b = inv * (X.T * y)
            e = y - X * b
            ve = (e.T * e)[0] / (n - (m + 2))
            vb = ve * inv[m + 1, m + 1]
            t = b[m + 1] / mp.sqrt(vb)
            h = mp.mpf(n - 1) / (mp.mpf(n - 1) + t ** 2)
            pv = mp.betainc(mp.mpf(n - 1) / 2, mp.mpf(1) / 2, 0, h, regularized=True)
            assert abs(got["beta"][l, j] - float(b[m + 1])) <= 1e-15 * abs(float(b[m + 1]))
            assert abs(got["var"][l, j] - float(vb)) <= 1e-15 * float(vb)
            assert abs(got["pval"][l, j] - float(pv)) <= 1e-15


def test_sym_eig_and_gp_ols_against_mpmath(exact):
    mp.mp.dps = 50
    rng = np.random.default_rng(6)
    n, p = 7, 40
    G = rng.uniform(0.05, 0.95, size=(p, n))
    K = exact.kinship(G)
    Km = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            Km[i, j] = sum(mp.mpf(float(G[l, i])) * mp.mpf(float(G[l, j])) for l in range(p)) / p
            assert abs(K[i, j] - float(Km[i, j])) <= 2e-16 * abs(float(Km[i, j]))
    ev, V = exact.sym_eig(K)
    E, Q = mp.eigsy(mp.matrix(K.tolist()))
    want = sorted((float(x) for x in E), reverse=True)
    assert np.allclose(ev, want, rtol=1e-14, atol=1e-17)
    assert np.allclose(V @ np.diag(ev) @ V.T, K, rtol=0, atol=1e-15 * abs(K).max())
    # gp::ols: b = X^T (X X^T)^-1 y on a training subset
    Xt = np.vstack([np.ones((1, n)), G])
    Y = rng.normal(size=(n, 2))
    idx = np.array([0, 2, 3, 5, 6])
    rc, b = exact.gp_ols(Xt, Y, idx)
    assert rc == 0
    Xs = mp.matrix(Xt[:, idx].T.tolist())
    for j in range(2):
        z = mp.lu_solve(Xs * Xs.T, mp.matrix(Y[idx, j].tolist()))
        want = Xs.T * z
        for c in range(p + 1):
            assert abs(b[c, j] - float(want[c])) <= 1e-14 * max(abs(float(want[c])), 1e-3)


def test_duplicated_pools_are_refused(exact):
    G = synth_G(300, 10, 3)
    G[:, 7] = G[:, 3]
    Xt = np.vstack([np.ones((1, 10)), G])
    rc, _ = exact.gp_ols(Xt, np.arange(10.0), np.arange(10))
    assert rc == -2


_CACHE = {}


@pytest.mark.parametrize("n,m", [(60, 1), (60, 3), (60, 8), (200, 1), (200, 3), (200, 8)])
def test_literal_oracle_error_on_covariate_fits_is_reported(oracle, exact, n, m, capsys):
    """What the 1e-6 tolerances of round 1 were hiding: the LITERAL normal equations (the oracle = the reference's
    operation order, gwas/ols.rs:58-118) against binary128 on the same inputs.  [1 | v_1 ...] is nearly collinear, so the
    literal route loses cond(X'X) * eps digits; the numbers are printed so that a reader sees which side a disagreement
    between GPU and oracle belongs to.  (The GPU side is asserted at 1e-10 in tests/test_gpu_exact.py.)"""
    import torch
    from poolgen_amd import synth
    if n not in _CACHE:   # one binary128 Jacobi per pool count: the eigenvectors do not depend on m
        G = synth_G(600, n, 17)
        _CACHE[n] = (G, synth.phenotypes(torch.from_numpy(G), n, k=1, seed=17), exact.kinship_covariates(G, force_m=8)[3])
    G, Y, C8 = _CACHE[n]
    C = C8[:, :m].copy()
    ex = exact.ols_covariate(G, Y, C)
    ref = oracle.ols_with_covariate(G, Y, covariate=C)
    rel = np.abs(ref["beta"] - ex["beta"]) / np.maximum(np.abs(ex["beta"]), 1e-300)
    err_b = float(np.max(np.minimum(rel, np.abs(ref["beta"] - ex["beta"]) / 1e-10 * 1e-10)))
    err_p = float(np.max(np.abs(ref["pval"] - ex["pval"])))
    X = np.column_stack([np.ones(n), C])
    cond = np.linalg.cond(X.T @ X)
    with capsys.disabled():
        print(f"\n[literal oracle vs binary128] n={n} m={m}: cond([1|C]'[1|C])={cond:.2e}  max rel|dbeta|={float(rel.max()):.2e}  max|dp|={err_p:.2e}")
    assert err_p < 1e-2 and np.isfinite(err_b)   # sanity only: the literal route is the noisy side, by how much is printed


def test_intercept_only_fits_agree_three_ways(oracle, exact):
    """m = 0 is well conditioned: literal oracle and binary128 agree far inside 1e-10."""
    n, p = 100, 800
    G = synth_G(p, n, 3)
    import torch
    from poolgen_amd import synth
    Y = synth.phenotypes(torch.from_numpy(G), n, k=2, seed=3)
    ex = exact.ols_covariate(G, Y, None)
    ref = oracle.ols_with_covariate(G, Y, force_m=0)
    assert np.allclose(ref["beta"], ex["beta"], rtol=1e-10, atol=1e-12)
    assert np.max(np.abs(ref["pval"] - ex["pval"])) < 1e-11


def test_gp_proxy_against_mpmath(oracle, exact):
    """The binary128 restatement of gp::ols_iterative_with_kinship_pca_covariate (gp/ols.rs:104-199), pinned against mpmath at 50
    digits on a case mpmath finishes in seconds: the quirky kinship (last locus left out, means over the FIRST n_rows pools),
    its leading eigenvector, the third coefficient of y ~ [1 | PC1 | x_j], the minimum-norm value for a constant locus -- and the
    literal oracle's distance from it, reported."""
    mp.mp.dps = 50
    n, P, k = 14, 9, 2
    rng = np.random.default_rng(8)
    Xt = np.vstack([np.ones((1, n)), rng.uniform(0.05, 0.95, size=(P - 1, n))])
    Xt[4, :] = 0.375                                   # a locus constant over the pools
    Y = rng.normal(size=(n, k))
    idx = np.array([i for i in range(n) if i % 5 != 2])
    nr = len(idx)
    got = exact.gp_proxy(Xt, Y, idx, n=n)
    # mpmath: the centred columns exactly as the reference forms them in fp64, everything after that at 50 digits
    pc = P - 1
    xc = np.empty((nr, pc))
    for j in range(pc):
        mean = 0.0
        for i_ in range(nr):
            mean += Xt[j, i_]
        mean = mean / nr
        xc[:, j] = Xt[j, idx] - mean
    A = mp.matrix(nr, nr)
    for a in range(nr):
        for c in range(nr):
            A[a, c] = mp.fsum(mp.mpf(float(xc[a, j])) * mp.mpf(float(xc[c, j])) for j in range(pc))
    E, Q = mp.eigsy(A)
    lead = max(range(nr), key=lambda i: E[i])
    e1 = [Q[a, lead] for a in range(nr)]
    for j_ in range(k):
        assert abs(got[0, j_] - float(mp.fsum(mp.mpf(float(Y[i, j_])) for i in idx) / nr)) <= 1e-15
    for j in range(1, P):
        X = mp.matrix(nr, 3)
        for a in range(nr):
            X[a, 0] = 1; X[a, 1] = e1[a]; X[a, 2] = mp.mpf(float(Xt[j, idx[a]]))
        for j_ in range(k):
            yv = mp.matrix([mp.mpf(float(Y[i, j_])) for i in idx])
            if j == 4:                                 # constant column: minimum-norm least squares
                X2 = X[:, 0:2]
                a01 = mp.lu_solve(X2.T * X2, X2.T * yv)
                c = mp.mpf(0.375)
                want = c * a01[0] / (1 + c * c)
            else:
                want = mp.lu_solve(X.T * X, X.T * yv)[2]
            assert abs(got[j, j_] - float(want)) <= 1e-13 * max(1.0, abs(float(want))), (j, j_, got[j, j_], float(want))
    lit = oracle.gp_proxy(Xt, Y, idx, n=n)
    assert np.allclose(lit, got, rtol=1e-7, atol=1e-9 * np.abs(got).max())
