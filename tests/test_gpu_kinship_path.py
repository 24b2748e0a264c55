"""GPU parity tests of the headline path (ols_iter_with_kinship) through the C ABI, checked
against the CPU oracle on identical inputs.  Tolerances (north_star): 1e-10 for coefficients
(relative) and p-values (absolute); kinship sums 1e-11 relative (summation order only)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-10
PTOL = 1e-10


def make(p, n, seed, device="cuda", ld=None):
    from poolgen_amd import synth
    G = synth.genotype_matrix(p, n, device, seed=seed, ld=ld)
    Y = synth.phenotypes(G, n, k=2, seed=seed)
    return G, Y


def cmp_fit(got, ref, what=""):
    beta, var, pv = (x.cpu().numpy() for x in got)
    ok = np.isfinite(ref["beta"])
    assert np.array_equal(np.isnan(beta), ~ok), what + " NaN pattern"
    # "within 1e-10": relative 1e-10, or absolute 1e-10 where the coefficient itself is ~0 (the
    # literal normal-equation oracle carries an ABSOLUTE error of ~cond*eps*|scale|; measured
    # against 80-bit arithmetic the GPU is the closer of the two, see DESIGN.md "Parity").
    # the floor is scale-free: 1e-10 of the largest coefficient (variance) of the run
    assert np.allclose(beta[ok], ref["beta"][ok], rtol=RTOL, atol=RTOL * float(np.max(np.abs(ref["beta"][ok]), initial=0.0))), what + " beta"
    assert np.allclose(var[ok], ref["var"][ok], rtol=RTOL, atol=RTOL * float(np.max(np.abs(ref["var"][ok]), initial=0.0))), what + " var"
    assert np.max(np.abs(pv[ok] - ref["pval"][ok])) <= PTOL, what + " pval"


@pytest.mark.parametrize("p,n,ld", [(3000, 200, None), (777, 100, None), (513, 5, None), (100, 37, 38),
                                    (64, 16, 16), (1, 8, 8), (1000, 250, None), (300, 209, 210),
                                    (700, 500, None), (333, 401, 402), (257, 385, 386), (200, 640, None),
                                    (20011, 300, None), (9001, 224, None), (1030, 360, 362), (40000, 257, 258)])   # 2-3 pool blocks: weighted pairs
def test_kinship_matches_oracle(engine, oracle, p, n, ld):
    G, _ = make(p, n, 11, ld=ld)
    S = engine.kinship_partial(G, n).cpu().numpy()
    K = oracle.kinship(G.cpu().numpy(), n)
    assert np.array_equal(S, S.T)
    assert np.allclose(S / p, K, rtol=1e-11, atol=0)


def test_gp_xxt_adds_the_intercept(engine):
    G, _ = make(500, 24, 5)
    S = engine.kinship_partial(G).cpu().numpy()
    X = engine.gp_xxt(G).cpu().numpy()
    assert np.allclose(X, S + 1.0, rtol=1e-15)


@pytest.mark.parametrize("p,n", [(5000, 200), (4099, 100), (130, 5), (1000, 63), (64, 33)])
def test_sweep_without_covariates(engine, oracle, p, n):
    G, Y = make(p, n, 3)
    engine.covariates_set(n, None, Y)
    got = engine.ols_sweep(G, 2, n)
    ref = oracle.ols_with_covariate(G.cpu().numpy(), Y, force_m=0, n=n)
    cmp_fit(got, ref, f"p={p} n={n}")


@pytest.mark.parametrize("p,n,ld,off", [(1, 8, 8, 0), (7, 5, 6, 0), (129, 37, 40, 0), (1000, 200, 208, 0), (4097, 100, 100, 0),
                                         (777, 200, 200, 3), (513, 100, 100, 1), (2049, 50, 50, 5), (300, 33, 34, 7), (640, 201, 202, 0),
                                         (1, 40, 40, 0), (63, 200, 200, 1), (65, 100, 124, 0), (17, 500, 500, 2), (4100, 68, 70, 0)])
def test_sweep_row_geometries(engine, oracle, p, n, ld, off):
    """The sweep kernels over awkward geometries (the matrix-core kernel from 33 pools up, the row kernel below): row pitches
    that put 1, 2, 4 or 8 loci into one 128-byte-aligned run (ld = 208 / 200 / 100 / 50, 6, 34), padding columns between n and ld
    (NaN there must never be read as data), odd n, pool counts that are not a multiple of the 8-pool chunk or of the load group,
    fewer loci than one 16-locus tile or one 64-locus group, a single locus, and a matrix that starts in the middle of an
    allocation (a rank's slab: not line-aligned)."""
    from poolgen_amd import synth
    Gfull = synth.genotype_matrix(p + off, n, "cuda", seed=77, ld=ld)
    if ld > n:
        Gfull[:, n:] = float("nan")                # padding must never be read as data: a NaN there would surface in the fits
    G = Gfull[off:]
    assert G.is_contiguous()
    Y = synth.phenotypes(Gfull, n, k=2, seed=77)
    engine.covariates_set(n, None, Y)
    got = engine.ols_sweep(G, 2, n)
    ref = oracle.ols_with_covariate(np.ascontiguousarray(G.cpu().numpy()[:, :n]), Y, force_m=0)
    cmp_fit(got, ref, f"p={p} n={n} ld={ld} off={off}")


@pytest.mark.parametrize("n,m,k", [(200, 1, 1), (200, 3, 2), (100, 8, 1), (40, 5, 3)])
def test_sweep_with_covariates(engine, oracle, exact, n, m, k):
    from test_gpu_exact import assert_close, formula_p
    p = 3000
    G, Y = make(p, n, 17)
    Y = Y[:, :1] if k == 1 else np.hstack([Y, Y[:, :1] ** 2])[:, :k]
    K = oracle.kinship(G.cpu().numpy(), n)
    w, V = np.linalg.eigh(K)
    C = V[:, ::-1][:, :m].copy()
    engine.covariates_set(n, C, Y)
    got = tuple(x.cpu().numpy() for x in engine.ols_sweep(G, k, n))
    # [1 | v1 ...] is nearly collinear (v1 of an uncentred kinship ~ 1/sqrt(n)): the literal normal-equation oracle loses
    # cond(X'X)*eps digits (7e-9 relative on beta, measured and printed by tests/test_gpu_exact.py), so the reference point is
    # the binary128 evaluation of the same cells, at north_star's 1e-10
    ex = exact.ols_covariate(G.cpu().numpy(), Y, C, n=n)
    assert_close(got, ex, formula_p(oracle, ex, n), f"n={n} m={m} k={k}")
    # and the literal oracle is within ITS OWN error of it (sanity of the oracle, not of the product)
    ref = oracle.ols_with_covariate(G.cpu().numpy(), Y, covariate=C, n=n)
    assert np.allclose(ref["beta"], ex["beta"], rtol=1e-6, atol=1e-10)


def test_full_path_default_threshold(engine, oracle):
    p, n = 20000, 200
    G, Y = make(p, n, 23)
    m, K, beta, var, pv = engine.ols_with_covariate(G, Y[:, :1], 0.75)
    ref = oracle.ols_with_covariate(G.cpu().numpy(), Y[:, :1], 0.75)
    assert m == ref["m"] == 0     # lambda_1 share of an uncentred kinship ~ 0.98 > 0.75
    assert np.allclose(K, ref["K"], rtol=1e-11)
    cmp_fit((beta, var, pv), ref, "full path")


@pytest.mark.parametrize("p", [1, 64, 130, 255, 257, 511, 777, 1025])
def test_fused_fits_stay_inside_their_outputs(engine, oracle, p):
    """The m = 0 fits closed from the fused sums (two loci per lane, k_sweep_finish_x2): locus counts around the workgroup size,
    outputs embedded in a sentinel-filled buffer that must come back untouched on both sides."""
    n = 40
    G, Y = make(p, n, 31)
    big = torch.full((3, p + 512, 1), -7.25, dtype=torch.float64, device="cuda")
    out = big[:, 256:256 + p, :]
    m, K, beta, var, pv = engine.ols_with_covariate(G, Y[:, :1], 0.0, force_m=0, out=out)   # (force_m = 0: the fused path whatever K says)
    assert m == 0
    torch.cuda.synchronize()
    assert bool((big[:, :256] == -7.25).all()) and bool((big[:, 256 + p:] == -7.25).all()), "wrote outside the output arrays"
    ref = oracle.ols_with_covariate(G.cpu().numpy(), Y[:, :1], force_m=0)
    cmp_fit((beta, var, pv), ref, f"fused p={p}")


@pytest.mark.parametrize("n,k,m", [(40, 1, 0), (40, 2, 3), (20, 1, 0), (200, 1, 8), (72, 3, 20)])
def test_sweep_stays_inside_its_outputs(engine, oracle, n, k, m):
    """Every sweep kernel (matrix-core: one trait, several traits, two accumulators; row kernel for short rows), locus counts around
    the 16-locus tile and the 64-locus group: the outputs sit in a sentinel-filled buffer that must come back untouched around them,
    and G is followed by NaNs that must not be read as data."""
    rng = np.random.default_rng(3)
    for p in (1, 15, 16, 17, 63, 64, 65, 127, 129, 1000):
        from poolgen_amd import synth
        Gbig = torch.full((p + 70, n), float("nan"), dtype=torch.float64, device="cuda")
        Gbig[:p] = synth.genotype_matrix(p, n, "cuda", seed=p)
        G = Gbig[:p]
        Y = synth.phenotypes(synth.genotype_matrix(500, n, "cuda", seed=5), n, k=2, seed=5)
        Y = np.hstack([Y, Y[:, :1] ** 2])[:, :k]
        C = None if m == 0 else np.linalg.qr(rng.normal(size=(n, m)))[0]
        engine.covariates_set(n, C, Y)
        big = torch.full((3, p + 512, k), -7.25, dtype=torch.float64, device="cuda")
        out = big[:, 256:256 + p, :]
        got = engine.ols_sweep(G, k, n, out)
        torch.cuda.synchronize()
        assert bool((big[:, :256] == -7.25).all()) and bool((big[:, 256 + p:] == -7.25).all()), f"p={p}: wrote outside the output arrays"
        ref = oracle.ols_with_covariate(G.cpu().numpy(), Y, covariate=C, n=n) if m else oracle.ols_with_covariate(G.cpu().numpy(), Y, force_m=0)
        cmp_fit(got, ref, f"n={n} k={k} m={m} p={p}")


def test_sweep_random_shapes(engine, oracle):
    """Forty seeded random shapes (pools 2 .. 300, row pitch n .. n + 40, 1-3 traits, 0-12 well-conditioned covariates, loci 1 .. 3000,
    slab offsets) through whichever sweep kernel the library picks, against the oracle."""
    from poolgen_amd import synth
    rng = np.random.default_rng(20261004)
    for case in range(40):
        n = int(rng.integers(5, 301))
        k = int(rng.integers(1, 4))
        m = int(rng.integers(0, min(12, n - 4) + 1)) if n >= 8 else 0
        ld = n + 2 * int(rng.integers(0, 21)); ld += ld & 1
        p = int(rng.integers(1, 3001))
        off = int(rng.integers(0, 9))
        Gfull = synth.genotype_matrix(p + off, n, "cuda", seed=1000 + case, ld=ld)
        if ld > n:
            Gfull[:, n:] = float("nan")
        G = Gfull[off:]
        Y = synth.phenotypes(synth.genotype_matrix(400, n, "cuda", seed=7), n, k=2, seed=case)
        Y = np.hstack([Y, Y[:, :1] ** 2])[:, :k]
        C = None if m == 0 else np.linalg.qr(rng.normal(size=(n, m)))[0]
        engine.covariates_set(n, C, Y)
        got = engine.ols_sweep(G, k, n)
        Gh = np.ascontiguousarray(G.cpu().numpy()[:, :n])
        ref = oracle.ols_with_covariate(Gh, Y, covariate=C, n=n) if m else oracle.ols_with_covariate(Gh, Y, force_m=0)
        cmp_fit(got, ref, f"case {case}: n={n} ld={ld} k={k} m={m} p={p} off={off}")

@pytest.mark.parametrize("forced", ["POOLGEN_SWEEP_V1", "POOLGEN_SWEEP_V2"])
def test_lane_per_locus_sweep_kernels_as_shipped(engine, oracle, forced, monkeypatch):
    """The default dispatch takes the matrix-core kernel for every shape the tests above use from 33 pools up; the two
    lane-per-locus kernels stay in the product for shapes it does not fit (more than ~1200 pools: the B table exceeds the
    LDS; more than 48 columns) and for short rows.  Forced through the library's own switch, both run the geometry cases
    (mid-chunk locus boundaries, paired stores, clamped prefetch, padding columns, slab offsets) and the random shapes."""
    monkeypatch.setenv(forced, "1")
    for args in [(1, 8, 8, 0), (7, 5, 6, 0), (129, 37, 40, 0), (1000, 200, 208, 0), (4097, 100, 100, 0), (777, 200, 200, 3),
                 (513, 100, 100, 1), (2049, 50, 50, 5), (300, 33, 34, 7), (640, 201, 202, 0), (63, 200, 200, 1), (65, 100, 124, 0),
                 (17, 500, 500, 2), (4100, 68, 70, 0)]:
        test_sweep_row_geometries(engine, oracle, *args)
    test_sweep_random_shapes(engine, oracle)



def test_full_path_rule_picks_covariates(engine, oracle, exact):
    from test_gpu_exact import assert_close, formula_p
    p, n = 4000, 60
    G, Y = make(p, n, 29)
    m, K, beta, var, pv = engine.ols_with_covariate(G, Y[:, :1], 0.99)
    ref = oracle.ols_with_covariate(G.cpu().numpy(), Y[:, :1], 0.99)
    assert m == ref["m"] and m >= 1
    ex = exact.ols_with_covariate(G.cpu().numpy(), Y[:, :1], 0.99)      # K -> eig -> rule -> fits in binary128
    assert ex["m"] == m
    # whole chain (own eigenvectors as covariates): the bound test_gpu_exact.py::test_full_path_against_binary128 explains
    assert_close((beta.cpu().numpy(), var.cpu().numpy(), pv.cpu().numpy()), ex, formula_p(oracle, ex, n), "rule picks covariates",
                 big_rtol=1e-7)


def test_degenerate_loci_are_nan_not_garbage(engine, oracle):
    n = 50
    G, Y = make(256, n, 31)
    G[5, :n] = 0.5          # exact zero pivot in the reference's LU (gwas/ols.rs:77-83) -> NaN
    G[77, :n] = 0.25
    engine.covariates_set(n, None, Y[:, :1])
    beta, var, pv = (x.cpu().numpy() for x in engine.ols_sweep(G, 1, n))
    ref = oracle.ols_with_covariate(G.cpu().numpy(), Y[:, :1], force_m=0)
    for l in (5, 77):
        assert np.isnan(ref["beta"][l, 0]) and np.isnan(beta[l, 0]) and np.isnan(pv[l, 0])
    ok = np.ones(256, bool); ok[[5, 77]] = False
    assert np.allclose(beta[ok], ref["beta"][ok], rtol=RTOL, atol=1e-10)


def test_planted_effect_and_special_cases(engine):
    n, p = 120, 640
    G, _ = make(p, n, 37)
    g = G[100, :n].cpu().numpy()
    y = 3.0 - 2.5 * g                       # exact linear function of locus 100
    engine.covariates_set(n, None, y)
    beta, var, pv = (x.cpu().numpy() for x in engine.ols_sweep(G, 1, n))
    assert abs(beta[100, 0] + 2.5) < 1e-9 and pv[100, 0] < 1e-12
    y0 = np.full(n, 7.0)                    # constant phenotype: beta = 0 -> t = 0 -> p = 1 (ols.rs:143-149)
    engine.covariates_set(n, None, y0)
    beta, var, pv = (x.cpu().numpy() for x in engine.ols_sweep(G, 1, n))
    assert np.all(np.abs(beta) < 1e-12) and np.all(pv == 1.0)


def test_rejects_bad_arguments(engine):
    from poolgen_amd import NativeError
    G, Y = make(128, 10, 41)
    Ybad = Y.copy(); Ybad[3, 0] = np.nan
    with pytest.raises(NativeError):
        engine.covariates_set(10, None, Ybad)
    with pytest.raises(NativeError):
        engine.kinship_set(engine.kinship_partial(G), 128, Y[:, :1], 1.0)   # m = n: no residual df


def test_full_size_properties(engine):
    """BASELINE config 3 shape (200 pools x 10M loci on one GPU): size-independent properties."""
    from poolgen_amd import synth
    n, p = 200, 10_000_000
    G = synth.genotype_matrix(p, n, "cuda")
    rng = np.random.default_rng(1)
    Y = rng.normal(size=(n, 2))
    S = engine.kinship_partial(G)
    # trace(S) = sum of squares of every entry; row sums of S = G^T (G 1)
    tr = float((G * G).sum())
    assert abs(float(torch.trace(S)) - tr) <= 1e-11 * tr
    rs = (G.sum(dim=1, keepdim=True) * G).sum(dim=0)
    assert torch.allclose(S.sum(dim=1), rs, rtol=1e-11, atol=0)
    # linearity of OLS in y: beta(y1 + y2) = beta(y1) + beta(y2); scale/shift invariance of p
    engine.covariates_set(n, None, Y)
    b2, v2, p2 = engine.ols_sweep(G, 2)
    engine.covariates_set(n, None, Y[:, 0] + Y[:, 1])
    b1, v1, p1 = engine.ols_sweep(G, 1)
    assert torch.allclose(b1[:, 0], b2[:, 0] + b2[:, 1], rtol=1e-9, atol=1e-12)
    engine.covariates_set(n, None, 10.0 - 4.0 * Y[:, :1])
    b3, v3, p3 = engine.ols_sweep(G, 1)
    assert torch.allclose(b3[:, 0], -4.0 * b2[:, 0], rtol=1e-9, atol=1e-12)
    assert float((p3[:, 0] - p2[:, 0]).abs().max()) < 1e-9
    assert bool(torch.isfinite(b2).all()) and float(p2.min()) >= 0.0 and float(p2.max()) <= 1.0
    # a checksum of checksums against torch on a slab
    sl = slice(5_000_000, 5_000_000 + 4096)
    g = G[sl]; y = torch.from_numpy(Y[:, 0]).cuda()
    gc = g - g.mean(dim=1, keepdim=True); yc = y - y.mean()
    bt = (gc @ yc) / (gc * gc).sum(dim=1)
    assert torch.allclose(b2[sl, 0], bt, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("p,n,k", [(20000, 200, 1), (7001, 100, 2), (513, 40, 3)])
def test_lazy_kinship_route(engine, oracle, p, n, k, monkeypatch):
    """pg_ols_kinship_dev with K_out = NULL: m = 0 is decided from lambda_1 / trace >= (1'K1 / n) / trace(K) (gwas/ols.rs:297-311
    returns 0 as soon as the leading share reaches x), two sums the intercept-only sweep forms on the side -- K is never built.
    Outputs bit-identical to the two-pass route (same sweep kernel), the oracle's at 1e-10; a matrix whose ones-vector bound does
    not clear x takes the full route and gives what the full route gives."""
    G, Y = make(p, n, 53)
    Y = np.hstack([Y, Y[:, :1] ** 2])[:, :k]
    m, K, beta, var, pv = engine.ols_with_covariate(G, Y, 0.75, want_K=False, n=n)
    torch.cuda.synchronize()
    assert m == 0 and K is None
    lazy = tuple(x.clone() for x in (beta, var, pv))
    engine.set_phenotypes(None)
    engine.covariates_set(n, None, Y)
    two = engine.ols_sweep(G, k, n)
    for a, b in zip(lazy, two):
        assert torch.equal(a, b), "lazy route vs the two-pass route: same kernel, same bits"
    ref = oracle.ols_with_covariate(G.cpu().numpy(), Y, 0.75, n=n)
    assert ref["m"] == 0
    cmp_fit(lazy, ref, "lazy route")
    # the switch: with it the full route runs and (m = 0) agrees at 1e-10 (fused sums: other arithmetic)
    monkeypatch.setenv("POOLGEN_NO_LAZY_KINSHIP", "1")
    m2, _, b2, v2, p2 = engine.ols_with_covariate(G, Y, 0.75, want_K=False, n=n)
    assert m2 == 0
    cmp_fit((b2, v2, p2), ref, "full route without K")
    monkeypatch.delenv("POOLGEN_NO_LAZY_KINSHIP")
    # a matrix the bound cannot decide: signed entries, the ones vector explains ~1/n of the trace
    Gs = torch.randn(4000, n + (n & 1), dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    Gs[:, n:] = 0.0
    ml, _, bl, vl, pl = engine.ols_with_covariate(Gs, Y, 0.2, want_K=False, n=n)
    mf, Kf, bf, vf, pf = engine.ols_with_covariate(Gs, Y, 0.2, want_K=True, n=n)
    assert ml == mf and ml >= 1
    for a, b in zip((bl, vl, pl), (bf, vf, pf)):
        assert torch.equal(a, b)


FULL_SLICES = ((0, 4096), (5_000_000 - 2048, 5_000_000 + 2048), (10_000_000 - 4096, 10_000_000))   # first, middle, last slab


def test_full_size_fused_headline_path(engine, oracle, monkeypatch):
    """The kernels bench.py TIMES, at the size it times them (BASELINE configs[2]: 200 pools x 10M loci, m = 0 at -x 0.75):
    k_kinship_syrk with the fused intercept-only sums + k_sweep_finish_x2, through engine.ols_with_covariate (= pg_ols_kinship_dev).
    Slab lengths, 32-bit descriptor offsets and the spec[l*NV+v] indexing depend on the size, so: three 4096-locus slices
    (first / middle / last slab) against the oracle at 1e-10, and every locus against the two-pass path (plain kinship +
    k_ols_sweep_mfma, itself oracle-checked on slabs by test_full_size_properties) at rtol 1e-9.  gwas/ols.rs:291-370."""
    from poolgen_amd import synth
    n, p = 200, 10_000_000
    G = synth.genotype_matrix(p, n, "cuda")
    Y = synth.phenotypes(G[:100000], n, k=1)
    monkeypatch.delenv("POOLGEN_TWO_PASS", raising=False)
    engine.profile(True); engine.profile_reset()
    m, K, beta, var, pv = engine.ols_with_covariate(G, Y, 0.75)
    torch.cuda.synchronize()
    fin_ms, fin_n = engine.profile_get("sweep_finish")
    sw_ms, sw_n = engine.profile_get("sweep")
    engine.profile(False)
    assert m == 0 and fin_n == 1 and sw_n == 0, "the fused path (k_sweep_finish, no sweep launch) is what must have run"
    fused = tuple(x.clone() for x in (beta, var, pv))
    # (a) slices against the oracle (m = 0: the fits need no kinship, so a slice is self-contained)
    for lo, hi in FULL_SLICES:
        ref = oracle.ols_with_covariate(G[lo:hi].cpu().numpy(), Y, force_m=0)
        cmp_fit(tuple(x[lo:hi] for x in fused), ref, f"fused slice [{lo}, {hi})")
    # K itself: trace and row sums (size-independent), then the two-pass kinship of the same matrix entry by entry
    engine.set_phenotypes(None)
    S2 = engine.kinship_partial(G)
    assert np.allclose(K, (S2 / p).cpu().numpy(), rtol=1e-12, atol=0), "fused and plain kinship passes disagree"
    # (b) every locus against the two-pass path
    engine.covariates_set(n, None, Y)
    b2, v2, p2 = engine.ols_sweep(G, 1)
    torch.cuda.synchronize()
    sb =float(b2.abs().max()); sv = float(v2.abs().max())
    assert torch.allclose(fused[0], b2, rtol=1e-9, atol=1e-10 * sb), "beta: fused vs two-pass"
    assert torch.allclose(fused[1], v2, rtol=1e-9, atol=1e-10 * sv), "var: fused vs two-pass"
    assert float((fused[2] - p2).abs().max()) <= 1e-10, "pval: fused vs two-pass"
    assert bool(torch.isfinite(fused[0]).all())


def test_full_size_forced_m8(engine, oracle, exact):
    """The m >= 1 step of the driver line (roofline_sweep.m8) at 200 x 10M: non-fused kinship, n x n eigen step, k_ols_sweep_mfma
    with [1 | C(8) | g].  The three slices against binary128 fits with the covariates the PRODUCT chose (handed to the arbiter,
    so this pins the sweep at full size at 1e-10); the covariates themselves against an independent eigen-decomposition of the
    product's K (subspace agreement); every locus of the one-call path against the explicit three-call path (identical kernels:
    bit-equal)."""
    from poolgen_amd import synth
    from test_gpu_exact import assert_close, formula_p
    n, p, m8 = 200, 10_000_000, 8
    G = synth.genotype_matrix(p, n, "cuda")
    Y = synth.phenotypes(G[:100000], n, k=1)
    m, K, beta, var, pv = engine.ols_with_covariate(G, Y, 0.75, force_m=m8)
    torch.cuda.synchronize()
    assert m == m8
    got = tuple(x.clone() for x in (beta, var, pv))
    # the explicit path: kinship_partial -> kinship_set -> ols_sweep
    engine.set_phenotypes(None)
    S = engine.kinship_partial(G)
    m2, K2, _ = engine.kinship_set(S, p, Y, 0.75, m8)
    b2, v2, p2 = engine.ols_sweep(G, 1)
    torch.cuda.synchronize()
    assert m2 == m8 and np.array_equal(K, K2)
    for a, b in zip(got, (b2, v2, p2)):
        assert torch.equal(a, b), "one-call and three-call paths run the same kernels on the same inputs"
    # covariates: the span of the 8 leading eigenvectors of K, from LAPACK (independent of pg_sym_eig_top)
    w, V = np.linalg.eigh(K)
    C = V[:, ::-1][:, :m8].copy()
    ev_host = np.empty(n); Vp = np.empty((n, m8))
    rc = engine._lib.pg_host_sym_eig_top(K.ctypes.data, n, m8, ev_host.ctypes.data, Vp.ctypes.data)
    assert rc == 0
    assert np.allclose(ev_host[:m8], w[::-1][:m8], rtol=1e-12)
    assert np.max(np.abs(np.abs(np.sum(Vp * C, axis=0)) - 1.0)) < 1e-9, "leading eigenvectors: product vs LAPACK"
    for lo, hi in FULL_SLICES:
        Gh = G[lo:hi].cpu().numpy()
        ex = exact.ols_covariate(Gh, Y, Vp, n=n)                    # binary128 cells with the product's own covariates
        assert_close(tuple(x[lo:hi].cpu().numpy() for x in got), ex, formula_p(oracle, ex, n), f"m8 slice [{lo}, {hi})")


@pytest.mark.parametrize("n,p,k,rows", [(24, 3000, 1, None), (60, 5000, 2, "odd"), (200, 2500, 3, "fold")])
def test_gp_ols_matches_oracle(engine, oracle, n, p, k, rows):
    """gp::ols (gp/ols.rs:47-72): b = X^T pinv(X X^T) y on a training subset of the pools."""
    G, Y = make(p, n, 43)
    Y = np.hstack([Y, Y[:, :1] * 0.5 + 1.0])[:, :k]
    idx = np.arange(n) if rows is None else (np.arange(1, n, 2) if rows == "odd" else np.array([i for i in range(n) if i % 10 != 3]))
    beta = engine.gp_ols(G, Y, idx, n=n).cpu().numpy()
    Xt = np.vstack([np.ones((1, n)), G.cpu().numpy()[:, :n]])
    rc, ref = oracle.gp_ols(Xt, Y, idx, n=n)
    assert rc == 0
    scale = np.abs(ref).max()
    assert np.allclose(beta, ref, rtol=1e-10, atol=1e-11 * scale)   # cond(X X^T) ~ 1e3..1e4: both sides within 1e-12 of binary128 (test_gpu_exact.py)
    # the defining property tested by the reference (gp/ols.rs:245-246): the training rows are fitted
    yhat = Xt.T[idx] @ beta
    assert np.allclose(yhat, Y[idx], atol=1e-6 * np.abs(Y).max())
    # and the precomputed full-data X X^T gives the same answer (principal sub-block reuse)
    beta2 = engine.gp_ols(G, Y, idx, XXt=engine.gp_xxt(G, n).cpu().numpy(), n=n).cpu().numpy()
    assert np.array_equal(beta, beta2)


def test_gp_ols_tall_design(engine, oracle):
    """The other branch of gp::ols (gp/ols.rs:72-99, x.nrows() >= x.ncols()): b = pinv(X'X) X'y.  The reference's own test of
    it (gp/ols.rs:208-246: 5 pools x (1 + 2) columns, the fit reproduces y to 4 decimals), then random tall designs against the
    oracle -- whole and on a training subset, one with a duplicated column (pseudo-inverse)."""
    y = (np.arange(1, 6) / 5.0).reshape(5, 1)
    Gt = (np.arange(1, 31, 3) / 30.0).reshape(5, 2).T.copy()            # 2 loci x 5 pools; ld must be even
    G = torch.zeros((2, 6), dtype=torch.float64, device="cuda"); G[:, :5] = torch.from_numpy(Gt).cuda()
    b = engine.gp_ols(G, y, np.arange(5), n=5).cpu().numpy()
    Xt = np.vstack([np.ones((1, 5)), Gt])
    assert [oracle.lib.orc_sensible_round(float(v), 4) for v in Xt.T @ b[:, 0]] == y.ravel().tolist()
    rc, ref = oracle.gp_ols(Xt, y, np.arange(5), n=5)
    assert rc == 0 and np.allclose(b, ref, rtol=1e-10, atol=1e-11 * np.abs(ref).max())
    rng = np.random.default_rng(5)
    for n, p, k, dup in ((40, 7, 2, False), (64, 63, 1, False), (30, 5, 3, True)):
        Gh = rng.random((p, n))
        if dup:
            Gh[3] = Gh[1]
        Y = rng.normal(size=(n, k))
        Gd = torch.from_numpy(Gh).cuda()
        Xt = np.vstack([np.ones((1, n)), Gh])
        for idx in (np.arange(n), np.arange(0, n, 2) if p + 1 <= n // 2 else np.arange(n)):
            got = engine.gp_ols(Gd, Y, idx, n=n).cpu().numpy()
            rc, ref = oracle.gp_ols(Xt, Y, idx, n=n)
            assert rc == 0
            assert np.allclose(got, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max()), (n, p, k, dup)


@pytest.mark.parametrize("n,p,k", [(100, 70001, 2), (37, 513, 1), (200, 20000, 4)])
def test_gp_predict_is_x_times_beta(engine, n, p, k):
    """The prediction step of the CV harness (multiply_views_xx, gp/cv.rs:160-168): yhat = [1 | G^T] beta."""
    G, _ = make(p, n, 61)
    rng = np.random.default_rng(3)
    beta = rng.normal(size=(1 + p, k))
    got = engine.gp_predict(G, torch.from_numpy(beta).cuda(), n=n)
    want = beta[0] + G.cpu().numpy()[:, :n].T @ beta[1:]
    assert np.allclose(got, want, rtol=1e-11, atol=1e-11 * np.abs(want).max())


def test_gp_ridge_leftover_group_only_trains(engine, oracle):
    """k_split's left-over group (penalise.rs:444-448, fold id == n_folds) is in every training set and in no
    validation set."""
    n, p = 48, 1500
    G, Y = make(p, n, 49)
    Y = Y[:, :1]
    rng = np.random.default_rng(9)
    rows = np.arange(n)
    n_folds, n_reps = 4, 2
    folds = np.stack([rng.permutation(np.concatenate([np.arange(40) % n_folds, np.full(8, n_folds)])) for _ in range(n_reps)])
    beta, lam, perf = engine.gp_ridge(G, Y, rows, folds, n_folds, alpha=0.0, n=n)
    Xt = np.vstack([np.ones((1, n)), G.cpu().numpy()[:, :n]])
    rb, rl, rp = oracle.penalised_lambda_path(Xt, Y, rows, folds, n_folds, alpha=0.0, n=n)
    assert np.array_equal(lam, rl)
    assert np.allclose(perf, rp, rtol=1e-10, atol=2.6e-8) and (np.abs(perf - rp) > 1e-10).mean() < 0.02   # 7-dp rounded r inside
    assert np.allclose(beta.cpu().numpy(), rb, rtol=1e-10, atol=1e-11 * np.abs(rb).max())


@pytest.mark.parametrize("n,p,k,rows", [(40, 3000, 2, None), (50, 2001, 1, "odd"), (64, 1500, 2, "drop")])
def test_gp_proxy_matches_oracle(engine, oracle, exact, n, p, k, rows):
    """ols_iterative_with_kinship_pca_covariate (gp/ols.rs:104-199): per-locus coefficient of y ~ [1 | PC1 | g] on the
    training pools, with the reference's kinship (last locus left out, means over the first n_rows pools).  The GPU is held to
    1e-10 against the binary128 restatement (oracle/poolgen_exact.c::exq_gp_proxy, pinned to mpmath); the literal oracle only has
    to be within ITS error of it."""
    G, Y = make(p, n, 71)
    Y = np.hstack([Y, Y[:, :1] * 0.5 + 1.0])[:, :k]
    idx = np.arange(n) if rows is None else (np.arange(1, n, 2) if rows == "odd" else np.array([i for i in range(n) if i % 10 != 3]))
    G[17, :] = 0.25                          # a locus constant over the pools: the minimum-norm branch of least_squares
    got = engine.gp_proxy(G, Y, idx, n=n).cpu().numpy()
    Xt = np.vstack([np.ones((1, n)), G.cpu().numpy()[:, :n]])
    ref = oracle.gp_proxy(Xt, Y, idx, n=n)
    assert not np.isnan(ref).any()
    ex = exact.gp_proxy(Xt, Y, idx, n=n)
    assert np.allclose(got, ex, rtol=1e-10, atol=1e-10 * np.abs(ex).max())
    assert np.allclose(ref, ex, rtol=1e-7, atol=1e-9 * np.abs(ex).max())      # the literal oracle's own error
    # and against numpy's own least squares for a few loci (the restated kinship included)
    nr = len(idx)
    xc = Xt[:-1, :].T[idx] - Xt[:-1, :nr].mean(axis=1)
    w, V = np.linalg.eigh(xc @ xc.T)
    ev = V[:, -1]
    for l in (1, 18, p // 2, p):
        sol = np.linalg.lstsq(np.column_stack([np.ones(nr), ev, Xt[l, idx]]), Y[idx], rcond=None)[0][2]
        assert np.allclose(got[l], sol, rtol=1e-6, atol=1e-8 * np.abs(ref).max()), l


@pytest.mark.parametrize("n,p,k,alpha,proxy", [(48, 2000, 1, -0.1, False), (40, 1500, 2, -0.1, False),
                                                (48, 2000, 1, 1.0, True), (44, 1200, 2, 0.0, True)])
def test_gp_penalised_family_matches_oracle(engine, oracle, exact, n, p, k, alpha, proxy):
    """penalise_glmnet (alpha < 0: the alpha x lambda grid, gp/penalise.rs:168-195, :479-498) and the
    *_with_iterative_proxy_norms models (:197-246) with explicit folds.  For the proxy models the oracle's path takes its fold fits
    and its proxy coefficients from the binary128 restatements (everything downstream -- expand_and_contract, error_index, the
    arg-min / mode rules -- stays the oracle's): 1e-10 throughout."""
    G, Y = make(p, n, 83)
    Y = Y[:, :k]
    rng = np.random.default_rng(12)
    rows = np.array([i for i in range(n) if i % 11 != 5])
    n_folds, n_reps = 3, 3
    folds = np.stack([rng.permutation(np.arange(len(rows)) % n_folds) for _ in range(n_reps)])
    beta, al, lam, perf = engine.gp_penalised(G, Y, rows, folds, n_folds, alpha, proxy, n=n)
    Xt = np.vstack([np.ones((1, n)), G.cpu().numpy()[:, :n]])
    if proxy:
        exact.install_into_oracle(oracle, True)
    try:
        rb, ra, rl, rp = oracle.penalised_path_general(Xt, Y, rows, folds, n_folds, alpha, proxy, n=n)
    finally:
        if proxy:
            exact.install_into_oracle(oracle, False)
    assert perf.shape == rp.shape == (n_reps, n_folds, 11 if alpha < 0 else 1, 11, k)
    ptol = 1e-10
    assert np.allclose(perf, rp, rtol=ptol, atol=2.6e-8) and (np.abs(perf - rp) > 1e-10).mean() < 0.02   # (7-dp rounded r inside the metrics)
    assert np.array_equal(al, ra) and np.array_equal(lam, rl)
    assert np.allclose(beta.cpu().numpy(), rb, rtol=ptol, atol=1e-10 * np.abs(rb).max())
    if alpha >= 0 and not proxy:
        return
    # the per-fold route (one pair of passes over G per fold) must agree with the fused one
    os.environ["POOLGEN_RIDGE_PER_FOLD"] = "1"
    try:
        beta2, al2, lam2, perf2 = engine.gp_penalised(G, Y, rows, folds, n_folds, alpha, proxy, n=n)
    finally:
        del os.environ["POOLGEN_RIDGE_PER_FOLD"]
    assert np.allclose(perf2, perf, rtol=1e-9, atol=1e-12) and np.array_equal(lam2, lam) and np.array_equal(al2, al)
    os.environ["POOLGEN_RIDGE_PER_REP"] = "1"   # one coefficient pass per repetition instead of 16 columns per pass across repetitions
    try:
        beta4, al4, lam4, perf4 = engine.gp_penalised(G, Y, rows, folds, n_folds, alpha, proxy, n=n)
    finally:
        del os.environ["POOLGEN_RIDGE_PER_REP"]
    assert np.array_equal(perf4, perf) and np.array_equal(lam4, lam) and np.array_equal(al4, al) and np.array_equal(beta4.cpu().numpy(), beta.cpu().numpy())
    # and the caller's own X X^T (the CV harness computes it once for all its fits) changes nothing
    beta3, al3, lam3, perf3 = engine.gp_penalised(G, Y, rows, folds, n_folds, alpha, proxy, n=n, XXt=engine.gp_xxt(G, n).cpu().numpy())
    assert np.array_equal(perf3, perf) and np.array_equal(beta3.cpu().numpy(), beta.cpu().numpy())


@pytest.mark.parametrize("n,p,k,n_folds", [(80, 1500, 2, 10), (40, 900, 2, 10), (120, 2500, 1, 16), (300, 4000, 1, 10)])
def test_gp_ridge_many_fold_columns(engine, oracle, n, p, k, n_folds):
    """More fold x trait columns than one MFMA tile holds (20 -> the 24-column VALU forms, Z in LDS or in the scalar
    cache), exactly 16, and the 10-column MFMA form at a pool count with a ragged last chunk."""
    G, Y = make(p, n, 91)
    Y = np.hstack([Y, Y[:, :1] * 0.3 - 1.0])[:, :k]
    rng = np.random.default_rng(13)
    rows = np.arange(n)
    folds = np.stack([rng.permutation(np.arange(n) % n_folds) for _ in range(2)])
    beta, lam, perf = engine.gp_ridge(G, Y, rows, folds, n_folds, alpha=0.0, n=n)
    Xt = np.vstack([np.ones((1, n)), G.cpu().numpy()[:, :n]])
    rb, rl, rp = oracle.penalised_lambda_path(Xt, Y, rows, folds, n_folds, alpha=0.0, n=n)
    assert np.allclose(perf, rp, rtol=1e-10, atol=2.6e-8) and (np.abs(perf - rp) > 1e-10).mean() < 0.02
    assert np.array_equal(lam, rl)
    assert np.allclose(beta.cpu().numpy(), rb, rtol=1e-10, atol=1e-11 * np.abs(rb).max())


def test_gp_ols_with_duplicated_pools_uses_the_pseudo_inverse(engine, oracle):
    """Two identical pools make X X^T singular: the reference's pinv (helpers.rs:463-482) drops the null
    direction.  The product's fast path (Cholesky) must hand such a matrix to the eigen-based pseudo-inverse."""
    n, p = 30, 2000
    G, Y = make(p, n, 51)
    G[:, 7] = G[:, 3]                      # pool 7 = pool 3
    Y = Y[:, :1].copy(); Y[7] = Y[3]
    idx = np.arange(n)
    beta = engine.gp_ols(G, Y, idx, n=n).cpu().numpy()
    Xt = np.vstack([np.ones((1, n)), G.cpu().numpy()[:, :n]])
    rc, ref = oracle.gp_ols(Xt, Y, idx, n=n)
    assert rc == 0
    assert np.allclose(beta, ref, rtol=1e-6, atol=1e-8 * np.abs(ref).max())
    assert np.allclose(Xt.T @ beta, Y, atol=1e-6 * np.abs(Y).max())


@pytest.mark.parametrize("n,p,k,alpha", [(60, 3000, 1, 0.0), (40, 2000, 2, 0.0), (50, 1500, 1, 1.0)])
def test_gp_ridge_path_matches_oracle(engine, oracle, n, p, k, alpha):
    """penalise_ridge_like / the lambda path with k-fold CV (gp/penalise.rs:133-159, :461-669) with the
    folds made explicit (the reference's are unseeded random, :452-453)."""
    G, Y = make(p, n, 47)
    Y = Y[:, :k]
    rng = np.random.default_rng(8)
    rows = np.array([i for i in range(n) if i % 9 != 4])          # an outer training subset
    n_folds, n_reps = 4, 3
    folds = np.stack([rng.permutation(np.arange(len(rows)) % n_folds) for _ in range(n_reps)])
    beta, lam, perf = engine.gp_ridge(G, Y, rows, folds, n_folds, alpha=alpha, n=n)
    Xt = np.vstack([np.ones((1, n)), G.cpu().numpy()[:, :n]])
    rb, rl, rp = oracle.penalised_lambda_path(Xt, Y, rows, folds, n_folds, alpha=alpha, n=n)
    # error indices: a 7-dp rounded correlation enters them (correlation_test.rs:70) -> 1e-7 grid / 4
    assert np.allclose(perf, rp, rtol=1e-10, atol=2.6e-8) and (np.abs(perf - rp) > 1e-10).mean() < 0.02
    assert np.array_equal(lam, rl)
    b = beta.cpu().numpy()
    assert np.allclose(b, rb, rtol=1e-10, atol=1e-11 * np.abs(rb).max())
    # the reference's unit vectors (gp/penalise.rs:709-720) through the same device code path: alpha = 1,
    # lambda = 0.5 contracts the small coefficients and moves their mass to the large ones
    assert len(np.unique(lam)) >= 1 and np.all((lam >= 0) & (lam <= 1))
    # all folds of a repetition share two passes over G (the default); one pair of passes per fold must agree
    import os
    os.environ["POOLGEN_RIDGE_PER_FOLD"] = "1"
    try:
        beta2, lam2, perf2 = engine.gp_ridge(G, Y, rows, folds, n_folds, alpha=alpha, n=n)
    finally:
        del os.environ["POOLGEN_RIDGE_PER_FOLD"]
    assert np.array_equal(lam2, lam) and np.allclose(perf2, perf, rtol=1e-12, atol=1e-13)
    assert np.allclose(beta2.cpu().numpy(), b, rtol=1e-12, atol=0)
    # round 4: the folds' columns of ALL repetitions and the all-rows fit are formed 16 per pass over G (here 3 x 4 x k + k columns:
    # batches that straddle repetitions and hold the final fit); one pass per repetition + one for the final fit must give the same bits
    os.environ["POOLGEN_RIDGE_PER_REP"] = "1"
    try:
        beta4, lam4, perf4 = engine.gp_ridge(G, Y, rows, folds, n_folds, alpha=alpha, n=n)
    finally:
        del os.environ["POOLGEN_RIDGE_PER_REP"]
    assert np.array_equal(lam4, lam) and np.array_equal(perf4, perf) and np.array_equal(beta4.cpu().numpy(), b)


@pytest.mark.parametrize("p,n,k", [(6000, 200, 2), (5000, 200, 3), (3000, 33, 1), (2000, 100, 2), (4000, 208, 1), (1500, 193, 1)])
def test_fused_and_two_pass_paths_agree_with_oracle(engine, oracle, p, n, k):
    """m = 0 through the fused kinship pass (k <= 2) or the two-pass path (k = 3), generic and 13-tile
    kernels, odd n: same answers as the oracle, and as each other."""
    G, Y2 = make(p, n, 53)
    Y = np.hstack([Y2, (Y2[:, :1] - Y2[:, 1:2]) ** 2])[:, :k]
    m, K, beta, var, pv = engine.ols_with_covariate(G, Y, 0.75, n=n)
    ref = oracle.ols_with_covariate(G.cpu().numpy(), Y, 0.75, n=n)
    assert m == ref["m"] == 0
    assert np.allclose(K, ref["K"], rtol=1e-11)
    cmp_fit((beta, var, pv), ref, f"fused/auto p={p} n={n} k={k}")
    # explicit two-pass path on the same inputs
    engine.set_phenotypes(None)
    S = engine.kinship_partial(G, n)
    m2, _, _ = engine.kinship_set(S, p, Y, 0.75)
    b2, v2, p2 = engine.ols_sweep(G, k, n)
    assert m2 == 0
    assert torch.allclose(b2, beta, rtol=1e-9, atol=1e-12) and float((p2 - pv).abs().max()) < 1e-11


def test_config4_full_size_properties(engine):
    """BASELINE config 4 shape (500 pools x 5 M loci, ridge-like path with 10-fold CV): size-independent properties.
    gp::ols interpolates its training pools (n < p: X b = y on the training rows, gp/ols.rs:245-246); lambda = 0 leaves
    the fit untouched, so the error index of the first path entry equals the one of plain OLS predictions; the fused and
    the per-fold route give the same error indices; the selected lambda is on the path; the penalised coefficients keep
    the intercept and the signs (expand_and_contract never crosses zero, gp/penalise.rs:296-313)."""
    from poolgen_amd import synth
    n, p = 500, 5_000_000
    G = synth.genotype_matrix(p, n, "cuda")
    Y = synth.phenotypes(G[:100000], n, k=1)
    rows = np.arange(n)
    folds = np.stack([(rows + r) % 10 for r in range(2)]).astype(np.int32)
    b_ols = engine.gp_ols(G, Y, rows, n=n)
    yhat = engine.gp_predict(G, b_ols, n=n)
    assert np.allclose(yhat, Y, atol=1e-6 * np.abs(Y).max())
    beta, lam, perf = engine.gp_ridge(G, Y, rows, folds, 10, alpha=0.0, n=n)
    assert perf.shape == (2, 10, 11, 1) and np.isfinite(perf).all() and lam[0] in [i / 10 for i in range(11)]
    b, bo = beta.cpu().numpy()[:, 0], b_ols.cpu().numpy()[:, 0]
    assert b[0] == bo[0]                                         # the intercept is never penalised (:356)
    assert np.all(b[1:] * bo[1:] >= 0.0)
    if lam[0] == 0.0:
        assert np.array_equal(b, bo)
    # lambda = 0: the held-out predictions are those of the fold's plain OLS fit
    tr = rows[folds[0] != 3]; va = rows[folds[0] == 3]
    yh = engine.gp_predict(G, engine.gp_ols(G, Y, tr, n=n), n=n)[va, 0]
    yt = Y[va, 0]
    rng_ = yt.max() - yt.min()
    d = yt - yh
    r = np.corrcoef(yt, yh)[0, 1]
    idx = ((1 - abs(round(r * 1e7) / 1e7)) + np.abs(d).sum() / rng_ + (d * d).sum() / rng_ ** 2 + np.sqrt((d * d).sum() / rng_ ** 2) / rng_) / 4
    assert abs(perf[0, 3, 0, 0] - idx) <= 1e-6 * max(1.0, abs(idx))
    os.environ["POOLGEN_RIDGE_PER_FOLD"] = "1"
    try:
        beta2, lam2, perf2 = engine.gp_ridge(G, Y, rows, folds[:1], 10, alpha=0.0, n=n)
    finally:
        del os.environ["POOLGEN_RIDGE_PER_FOLD"]
    assert np.allclose(perf2[0], perf[0], rtol=1e-7, atol=1e-10)
