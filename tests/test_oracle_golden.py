"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md section 8c).  CPU only."""
import json
import math
from pathlib import Path

import numpy as np

GOLD = Path(__file__).parent / "golden"
LIT = json.loads((GOLD / "reference_literals.json").read_text())


def six(counts_at):
    """n x 2 (A, T) counts -> n x 6 sync layout A,T,C,G,N,D."""
    c = np.zeros((len(counts_at), 6), dtype=np.uint64)
    c[:, :2] = np.asarray(counts_at, dtype=np.uint64)
    return c


def test_statrs_students_t_and_pearson_golden(oracle):
    g = LIT["pearson"]
    r, p = oracle.pearson(g["x"], g["y"])
    assert r == oracle.lib.orc_sensible_round(g["r_unrounded"], 7)  # correlation_test.rs:138,179
    assert p == g["pval"]                                            # correlation_test.rs:139,180 (bit-exact)


def test_correlation_csv_line_golden(oracle):
    g = LIT["pearson"]
    f = oracle.filt(maf=g["min_allele_frequency"])
    line = oracle.correlation_csv("Chromosome1", 12345, six(g["counts_AT"]), np.array(g["y"]).reshape(5, 1),
                                  g["pool_sizes"], f)
    assert line == g["line"]  # correlation_test.rs:140-141,181


def test_pearson_nan_cases(oracle):
    nan = float("nan")  # correlation_test.rs:182-205
    r, _ = oracle.pearson([0.1, 0.2, nan, nan, 0.5, 0.6], [0.1, 0.2, nan, nan, 0.5, 0.6])
    assert oracle.lib.orc_sensible_round(r, 2) == 1.00
    r, _ = oracle.pearson([0.1, 0.2, nan, nan, 0.5, 0.6], [0.1, 0.2, nan, 0.4, nan, 0.6])
    assert oracle.lib.orc_sensible_round(r, 2) == 1.00
    r, _ = oracle.pearson([nan, nan, nan], [nan, nan, nan])
    assert math.isnan(r)


def test_chisq_csv_line_golden(oracle):
    g = LIT["chisq"]
    f = oracle.filt(maf=g["min_allele_frequency"])
    line = oracle.chisq_csv("Chromosome1", 12345, six(g["counts_AT"]), g["pool_sizes"], f)
    assert line == g["line"]  # chisq_test.rs:57,81 (statrs ChiSquared cdf, bit-exact)


def test_sync_parse_filter_sort_golden(oracle):
    g = LIT["sync_line"]
    n, chrom, pos, counts = oracle.parse_sync_line(g["line"])
    assert (n, chrom, pos) == (5, "Chromosome1", 456527)
    assert counts.tolist() == g["counts"]                       # sync.rs:1435-1470, 1611
    fr = oracle.to_frequencies(counts)                           # sync.rs:1443-1460, 1612
    expect = np.asarray(g["counts"], dtype=float)
    expect = expect / expect.sum(axis=1, keepdims=True)
    assert np.array_equal(fr, expect)
    f = oracle.filt(maf=g["min_allele_frequency"])
    ids, fc = oracle.filter_locus(counts, g["pool_sizes"], f)
    assert "".join("ATCGND"[i] for i in ids) == g["filtered_alleles"]   # sync.rs:1480-1488, 1613
    assert fc.tolist() == [[row[1], row[2]] for row in g["counts"]]
    ffr = oracle.to_frequencies(fc)                              # sync.rs:1489-1498, 1614
    e4 = np.asarray(fc, dtype=float)
    e4 = e4 / e4.sum(axis=1, keepdims=True)
    assert np.array_equal(ffr, e4)
    sfr, sids = oracle.sort_by_allele_freq(ffr, ids, True)      # sync.rs:1499-1515, 1615
    assert "".join("ATCGND"[i] for i in sids) == g["sorted_alleles"]
    assert np.array_equal(sfr, e4[:, ::-1])


def test_comment_and_bad_lines(oracle):
    assert oracle.parse_sync_line("#chr\tpos\tref\tp1\n")[0] == 0      # sync.rs:111-114
    assert oracle.parse_sync_line("chr1\tabc\tC\t1:0:0:0:0:0\n")[0] < 0  # sync.rs:125-133
    n, chrom, pos, c = oracle.parse_sync_line("chr1\t7\tC\t1:2:3:4:5:6\r\n")  # sync.rs:105-110
    assert (n, chrom, pos, c.tolist()) == (1, "chr1", 7, [[1, 2, 3, 4, 5, 6]])


def test_first_locus_of_fixture_golden(oracle):
    g = LIT["loaded_first_locus"]
    lines = (GOLD / "test.sync").read_text().splitlines()
    n, chrom, pos, counts = oracle.parse_sync_line(lines[1])
    assert (chrom, pos) == (g["chromosome"], g["position"])
    f = oracle.filt(maf=0.005)
    ids, fc = oracle.filter_locus(counts, [20.0] * 5, f)
    fr, ids = oracle.sort_by_allele_freq(oracle.to_frequencies(fc), ids, True)
    # keep_p_minus_1: drop the most frequent allele (sync.rs:1033-1037)
    assert "".join("ATCGND"[i] for i in ids[1:]) == g["alleles"]
    assert fr[:, 1].tolist() == g["freq"]                       # sync.rs:1523-1531, 1616


def test_helpers_golden(oracle):
    h = LIT["helpers"]
    for x, nd, want in h["sensible_round"]:
        assert oracle.lib.orc_sensible_round(x, nd) == want        # helpers.rs:505
    for x, nd, want in h["roundup_own"]:
        assert oracle.round_own(x, nd) == want                      # helpers.rs:506-509
    a = np.arange(15, dtype=float).reshape(5, 3)
    b = a / 2.0
    w3 = np.array([1, 3, 4], dtype=np.int64); x2 = np.array([0, 2], dtype=np.int64)
    y2 = np.array([1, 3], dtype=np.int64); z2 = np.array([0, 1], dtype=np.int64)
    out = np.empty(6)
    oracle.lib.orc_multiply_views_xx(a.ctypes.data, 3, b.ctypes.data, 3, w3.ctypes.data, 3, x2.ctypes.data,
                                     y2.ctypes.data, 2, z2.ctypes.data, 2, out.ctypes.data)
    assert out.tolist() == h["mv_xx"]                                # helpers.rs:525-528
    out = np.empty(4)
    oracle.lib.orc_multiply_views_xtx(a.ctypes.data, 3, b.ctypes.data, 3, w3.ctypes.data, 3, x2.ctypes.data, 2,
                                      w3.ctypes.data, z2.ctypes.data, 2, out.ctypes.data)
    assert out.tolist() == h["mv_xtx"]                               # helpers.rs:529-532
    out = np.empty(9)
    oracle.lib.orc_multiply_views_xxt(a.ctypes.data, 3, b.ctypes.data, 3, w3.ctypes.data, 3, x2.ctypes.data, 2,
                                      w3.ctypes.data, 3, z2.ctypes.data, out.ctypes.data)
    assert out.tolist() == h["mv_xxt"]                               # helpers.rs:533-540
    v = np.array([float(x) for x in h["mean_ignore_nan"][0]])
    assert oracle.lib.orc_mean_ignore_nan(v.ctypes.data, 5, 1) == h["mean_ignore_nan"][1]  # helpers.rs:590-591


def test_rust_display_formatting(oracle):
    # Rust `{}` for f64: shortest round-trip, never an exponent (quirk 10 of SURVEY.md)
    cases = [(0.3, "0.3"), (4.0, "4"), (1e-7, "0.0000001"), (1.5e22, "15000000000000000000000"),
             (0.1 + 0.2, "0.30000000000000004"), (-0.0, "-0"), (0.0, "0"), (123456.789, "123456.789"),
             (5e-324, "0." + "0" * 323 + "5"), (-2.5, "-2.5"), (1e21, "1" + "0" * 21)]
    for x, want in cases:
        assert oracle.fmt(x) == want
    assert oracle.fmt(float("nan")) == "NaN" and oracle.fmt(float("inf")) == "inf"
    assert oracle.round_own(0.7778468292004548, 12) == "0.7778468292"
    assert oracle.round_own(2.0, 6) == "2"          # to_string shorter than n_digits -> returned as is


def test_ols_fit_commented_golden_beta(oracle):
    g = LIT["ols_commented_golden"]
    X = np.array(g["x"]); Y = np.array(g["y"])
    for j in range(2):
        rc, b, v, t, p = oracle.ols_fit(X, Y[:, j])
        assert rc == 0
        for i in (1, 2):
            assert oracle.round_own(b[i], 6) == oracle.fmt(g["beta_6dp"][i - 1][j])   # gwas/ols.rs:534
        # live code uses df = n - 1 (ols.rs:139); independent check with scipy
        from scipy import stats
        assert np.allclose(p, 2 * stats.t.sf(np.abs(t), 4), rtol=0, atol=1e-13)


def test_ols_fit_structural_like_reference(oracle):
    # mirrors gwas/ols.rs:447-525: Bernoulli X (100 x 51), 10 unit effects, noiseless y
    rng = np.random.default_rng(7)
    X = np.ones((100, 51)); X[:, 1:] = rng.integers(0, 2, size=(100, 50))
    b = np.zeros(51); idx = [1, 5, 10, 15, 20, 25, 30, 35, 40, 45]
    b[idx] = [1, -1, 1, -1, 1, -1, 1, -1, 1, -1]
    rc, bh, v, t, p = oracle.ols_fit(X, X @ b)
    assert rc == 0
    assert int(np.sum(np.abs(bh) > 1e-7)) == 10                       # ols.rs:524
    assert oracle.lib.orc_sensible_round(float(np.sum(p[idx])), 7) == 0.0  # ols.rs:525


def test_ols_fit_singular_is_error(oracle):
    X = np.ones((5, 2)); X[:, 1] = 0.5                               # constant column: exact zero pivot
    rc, *_ = oracle.ols_fit(X, np.arange(5.0))
    assert rc == -1                                                    # gwas/ols.rs:77-83


def test_gp_ols_fit_property(oracle):
    # gp/ols.rs:208-246: wide (5 x 10) and tall (5 x 3) designs reproduce y to 4 dp
    y = np.arange(1, 6) / 5.0
    wide = np.vstack([np.ones(5), (np.arange(1, 46) / 45.0).reshape(5, 9).T])       # (1+9) x 5, locus-major
    tall = np.vstack([np.ones(5), (np.arange(1, 31, 3) / 30.0).reshape(5, 2).T])    # (1+2) x 5
    for Xt in (wide, tall):
        rc, b = oracle.gp_ols(Xt, y, np.arange(5))
        assert rc == 0
        yhat = Xt.T @ b[:, 0]
        assert [oracle.lib.orc_sensible_round(float(v), 4) for v in yhat] == y.tolist()


def test_expand_and_contract_golden(oracle):
    g = LIT["expand_and_contract"]
    b = np.array(g["b"]).reshape(7, 1); c = np.array(g["c"]).reshape(7, 1)
    assert oracle.expand_and_contract(b, b, g["alpha"], g["lambda"]).ravel().tolist() == g["expected_b"]
    assert oracle.expand_and_contract(c, c, g["alpha"], g["lambda"]).ravel().tolist() == g["expected_c"]


def test_n_eigenvecs_rule_literal(oracle):
    # gwas/ols.rs:297-311: m = first i with cum[i] >= x (vectors explain < x); never checks the last
    assert oracle.n_eigenvecs([0.98, 0.01, 0.01], 0.75) == 0
    assert oracle.n_eigenvecs([0.5, 0.3, 0.1, 0.1], 0.75) == 1
    assert oracle.n_eigenvecs([0.4, 0.3, 0.2, 0.1], 0.85) == 2
    assert oracle.n_eigenvecs([0.4, 0.3, 0.2, 0.1], 0.95) == 4     # cum[n-1] is never tested (loop bound)
    assert oracle.n_eigenvecs([0.4, 0.3, 0.2, 0.1], 1.0) == 4     # threshold never reached -> n


def test_sym_eig_against_numpy(oracle):
    rng = np.random.default_rng(3)
    A = rng.normal(size=(40, 40)); A = A @ A.T
    ev, V = oracle.sym_eig(A)
    w = np.linalg.eigvalsh(A)[::-1]
    assert np.allclose(ev, w, rtol=1e-12)
    assert np.allclose(A @ V, V * ev, atol=1e-10 * w[0])


def test_pileup_reference_literal(oracle):
    """base/pileup.rs:553-659: parse (indels, read starts/ends, deletions), counts in A,T,C,G,D,N order, and the
    phred filter that turns the Q = 22 read of pool 5 into an N and removes it."""
    g = LIT["pileup"]
    ps = g["filter"]["pool_sizes"]
    rc, text = oracle.pileup_to_sync(g["line"], ps, remove_ns=True, max_base_error_rate=1.0, min_allele_frequency=0.0)
    want = "Chromosome1\t456527\tC\t" + "\t".join(":".join(map(str, r)) for r in g["counts_ATCGDN"]) + "\n"
    assert rc == len(want) and text == want
    f = g["filter"]
    rc, text = oracle.pileup_to_sync(g["line"], ps, f["remove_ns"], f["max_base_error_rate"], f["min_coverage_depth"],
                                     f["min_coverage_breadth"], f["min_allele_frequency"])
    cov = [sum(map(int, fld.split(":"))) for fld in text.rstrip("\n").split("\t")[3:]]
    assert rc > 0 and cov == g["filtered_coverages"]
    assert text.rstrip("\n").split("\t")[-1] == "0:1:5:0:0:0"
    # the reference's min-allele-frequency loop only ever tests column 1 (T) first: a locus without T reads is dropped
    rc, _ = oracle.pileup_to_sync("c\t1\tA\t2\t.C\tJJ\t2\t..\tJJ", [0.5, 0.5], min_allele_frequency=0.01)
    assert rc == 0
    rc, _ = oracle.pileup_to_sync("c\t1\tA\t2\t.T\tJJ\t2\t..\tJJ", [0.5, 0.5], min_allele_frequency=0.01)
    assert rc > 0
    # malformed lines are fatal in the reference (lparse under .expect())
    assert oracle.pileup_to_sync("c\tx\tA\t1\t.\tJ", [1.0])[0] == -1
    assert oracle.pileup_to_sync("c\t1\tA\t2\t.\tJ", [1.0])[0] == -4
    assert oracle.pileup_to_sync("c\t1\tA\t1\t.+xa\tJ", [1.0])[0] == -5


def _popgen_case(oracle, key):
    g = LIT["popgen"]
    x = np.array(g[key]["x_pool_by_column"])
    idx, lc, lp = oracle.count_loci(g["labels"]["chromosome"], g["labels"]["position"])
    cov = np.array(g["coverages_pool_by_locus"]).T                   # oracle: locus x pool
    wh, wt = oracle.sliding_windows(lc[:-1], lp[:-1], g[key]["window_size_bp"], g[key]["window_slide_size_bp"],
                                    g[key]["min_loci_per_window"])
    return x.T.copy(), idx, cov, wh, wt


def test_sliding_windows_golden(oracle):
    for c in LIT["popgen"]["windows"]:                               # helpers.rs:541-583
        h, t = oracle.sliding_windows(c["chr"], c["pos"], c["window_size_bp"], c["window_slide_size_bp"], c["min_loci_per_window"])
        assert h.tolist() == c["head"] and t.tolist() == c["tail"]


def test_fst_golden(oracle):
    Xt, idx, cov, wh, wt = _popgen_case(oracle, "fst")               # popgen/fst.rs:262-356
    rc, mean, win = oracle.fst(Xt, idx, cov, wh, wt)
    assert rc == 0
    e = LIT["popgen"]["fst"]["expect"]
    printed = np.array([[float(oracle.round_own(v, 8)) for v in row] for row in mean])   # as written to the file (:146-155)
    assert np.all(np.diag(printed) == e["diag"])
    assert printed[0, 1] == e["pop1_2"] and printed[1, 0] == e["pop2_1"]
    assert printed[3, 4] == e["pop4_5"] and printed[4, 3] == e["pop5_4"]
    assert abs(printed[0, 2] - e["pop1_3_about"]) < e["about_tol"] and abs(printed[2, 1] - e["pop3_2_about"]) < e["about_tol"]
    assert win.shape == (2, 25) and np.array_equal(win.reshape(2, 5, 5), win.reshape(2, 5, 5).transpose(0, 2, 1))
    # frequencies that do not sum to one trip the reference's assert (:66)
    bad = Xt.copy(); bad[1, 0] = 0.9
    assert oracle.fst(bad, idx, cov, wh, wt)[0] == -1


def test_pi_golden(oracle):
    Xt, idx, cov, wh, wt = _popgen_case(oracle, "pi")                # popgen/pi.rs:202-297
    win, mean = oracle.theta_pi(Xt, idx, cov, wh, wt)
    e = LIT["popgen"]["pi"]["expect_round4"]
    assert oracle.round_own(win[0, 1], 4) == e["pop2_window1"] and oracle.round_own(win[1, 1], 4) == e["pop2_window2"]
    assert oracle.round_own(win[0, 4], 4) == e["pop5_window1"] and oracle.round_own(win[1, 4], 4) == e["pop5_window2"]
    assert np.allclose(mean, win.mean(axis=0), rtol=1e-15)


def test_mle_restatement_sits_at_the_analytic_optimum(oracle):
    """gwas/mle.rs is PARITY UNPINNED (no reference test, solver source absent): the restated Nelder-Mead must at least find the
    optimum of the reference's own cost function -- the OLS coefficients, sigma^2 = 2 RSS / n (mle.rs:27 has 1 / sigma^2)."""
    rng = np.random.default_rng(3)
    n = 40
    X = np.column_stack([np.ones(n), rng.random(n)])
    y = 1.0 + 2.5 * X[:, 1] + 0.3 * rng.normal(size=n)
    rc, b, v, t, p = oracle.mle_fit(X, y)
    assert rc == 0
    bo = np.linalg.lstsq(X, y, rcond=None)[0]
    rss = float(((y - X @ bo) ** 2).sum())
    assert np.allclose(b, bo, rtol=0, atol=2e-5)
    ve = 2.0 * rss / n
    vopt = ve * np.diag(np.linalg.inv(X.T @ X))
    assert np.allclose(v, vopt, rtol=1e-3)
    assert np.allclose(t, b / v)                                  # as written (mle.rs:175)
