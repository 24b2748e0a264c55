"""GPU parity of the sync-derived per-locus operators (ols_iter, pearson_corr, chisq_test) through
the C ABI against the CPU oracle: the reference's own fixture (BASELINE config 1) and synthetic
batches.  Index work (which loci/alleles are emitted, allele order) must be bit-exact; mean
frequencies are bit-exact; statistics within 1e-10."""
from pathlib import Path

import numpy as np
import pytest
import torch

import rustfmt

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
AL = "ATCGND"


@pytest.fixture(params=["rows", "stream"])
def ols_kernel(request, monkeypatch):
    """ols_iter from 32 pools up has two first-pass kernels: the order-free one (a locus per row of lanes; the robust default) and
    the lane-per-locus streaming pass (pool-order sums; what a context switches to after a clean batch).  Left alone the library
    picks by what the previous batch looked like; the tests that use this fixture run under both, fixed."""
    monkeypatch.setenv("POOLGEN_OLS_ITER_KERNEL", request.param)
    return request.param


def load_fixture(oracle):
    rows = []
    for line in (GOLD / "test.sync").read_text().splitlines():
        n, chrom, pos, counts = oracle.parse_sync_line(line)
        if n > 0:
            rows.append((chrom, pos, counts))
    Y = np.loadtxt(GOLD / "test.csv", delimiter=",", comments="#", usecols=(2, 3))
    return rows, Y, np.full(5, 20.0)


def to_dev(rows):
    c = np.stack([r[2] for r in rows]).astype(np.int32)
    return torch.from_numpy(c).cuda()


def flt_pair(oracle, **kw):
    from poolgen_amd import Filter
    f = Filter(remove_ns=kw.get("remove_ns", True), min_coverage_depth=kw.get("min_cov", 1),
               min_allele_frequency=kw.get("maf", 0.001), max_missingness_rate=kw.get("miss", 0.0))
    return f, oracle.filt(kw.get("remove_ns", True), kw.get("min_cov", 1), kw.get("maf", 0.001), kw.get("miss", 0.0))


def design_cond(oracle, counts, ps, fo):
    """cond_2 of the reference's design matrix [1 | sorted frequencies without the major allele]."""
    ids, fc = oracle.filter_locus(counts, ps, fo)
    fr, ids = oracle.sort_by_allele_freq(oracle.to_frequencies(fc), ids, True)
    X = np.ones_like(fr); X[:, 1:] = fr[:, 1:]
    return np.linalg.cond(X)


def check_stat_op(gpu, ref_fn, rows_counts, Y, ps, fo, stat_rtol=1e-10, stat_atol=1e-10, oracle=None, mf_rtol=None):
    """Index work bit-exact for every locus; statistics within tolerance for every locus whose
    design is numerically full rank.  Rank-deficient designs (duplicated pools/alleles: cond(X) >
    1e7, i.e. cond(X'X) > 1e14) make the reference print rounding noise (negative variances,
    p = 1): there only the emission pattern is comparable, and the count of such loci is reported."""
    n_out, ids, mf, stat, pv = (x.cpu().numpy() for x in gpu)
    if mf_rtol is None:
        # ols_iter from 32 pools up runs the order-free kernel (k_ols_rows): its mean frequency is the reference's to 1e-12 (the
        # reference prints 8 decimals of it, gwas/ols.rs:266-269), everything else -- and every other operator -- bit for bit
        mf_rtol = 1e-12 if (ref_fn.__name__ == "ols_iterate_locus" and np.asarray(rows_counts[0]).shape[0] >= 32) else 0.0
    degenerate = 0
    for l, counts in enumerate(rows_counts):
        na, rid, rmf, rs, rp = ref_fn(counts, Y, ps, fo)
        na = max(na, 0)
        assert n_out[l] == na, f"locus {l}: emitted {n_out[l]} alleles, oracle {na}"
        if na == 0:
            continue
        assert ids[l, :na].tolist() == rid, f"locus {l}: allele order"
        if mf_rtol == 0.0:
            assert np.array_equal(mf[l, :na], np.asarray(rmf), equal_nan=True), f"locus {l}: mean frequency not bit-exact"
        else:
            assert np.allclose(mf[l, :na], np.asarray(rmf), rtol=mf_rtol, atol=0, equal_nan=True), f"locus {l}: mean frequency"
        g, r = stat[l, :na], rs
        err = np.abs(g - r) - stat_rtol * np.abs(r)
        err[np.isnan(g) & np.isnan(r)] = 0.0
        pe = np.abs(pv[l, :na] - rp)
        pe[np.isnan(pv[l, :na]) & np.isnan(rp)] = 0.0
        if not (np.all(err <= stat_atol) and np.all(pe <= 1e-10)):
            if oracle is not None and design_cond(oracle, counts, ps, fo) > 1e7:
                degenerate += 1
                continue
            raise AssertionError(f"locus {l}: stat {g} vs {r}; pval {pv[l, :na]} vs {rp}")
    assert degenerate <= max(2, len(rows_counts) // 50), f"{degenerate} rank-deficient loci"
    return degenerate


@pytest.mark.parametrize("kw", [dict(), dict(min_cov=10, maf=0.01)])   # the two CI invocations, rust.yml:36-37
def test_ols_iter_on_reference_fixture(engine, oracle, kw):
    rows, Y, ps = load_fixture(oracle)
    f, fo = flt_pair(oracle, **kw)
    gpu = engine.ols_iterate(to_dev(rows), ps, f, Y)
    check_stat_op(gpu, oracle.ols_iterate_locus, [r[2] for r in rows], Y, ps, fo, oracle=oracle)
    # CSV text (ols.rs:255-275) rebuilt from the GPU numbers must equal the oracle's lines
    n_out, ids, mf, stat, pv = (x.cpu().numpy() for x in gpu)
    bad = 0
    total = 0
    for l, (chrom, pos, counts) in enumerate(rows):
        want = oracle.ols_iterate_csv(chrom, pos, counts, Y, ps, fo) or ""
        got = ""
        for i in range(n_out[l]):
            for j in range(Y.shape[1]):
                got += ",".join([chrom, str(pos), AL[ids[l, i]], rustfmt.roundup_own(mf[l, i], 8), f"Pheno_{j}",
                                 rustfmt.roundup_own(stat[l, i, j], 6), rustfmt.roundup_own(pv[l, i, j], 12)]) + "\n"
        total += want.count("\n")
        bad += sum(1 for a, b in zip(got.splitlines(), want.splitlines()) if a != b) + abs(got.count("\n") - want.count("\n"))
    assert total > 4000
    # rows of rank-deficient loci print noise on both sides (see check_stat_op); everything else is identical text
    assert bad <= max(1, total // 100), f"{bad} of {total} CSV rows differ in a printed digit"
    print(f"ols_iter CSV: {total - bad} of {total} rows textually identical")


@pytest.mark.parametrize("kw", [dict(), dict(min_cov=10, maf=0.01)])   # rust.yml:34-35
def test_pearson_on_reference_fixture(engine, oracle, kw):
    rows, Y, ps = load_fixture(oracle)
    f, fo = flt_pair(oracle, **kw)
    gpu = engine.correlation(to_dev(rows), ps, f, Y)
    # r is rounded to 7 dp by the reference (correlation_test.rs:70): allow one unit of that grid
    check_stat_op(gpu, oracle.correlation_locus, [r[2] for r in rows], Y, ps, fo, stat_rtol=0, stat_atol=1.0000001e-7)


@pytest.mark.parametrize("kw", [dict(), dict(min_cov=10, maf=0.01)])   # rust.yml:32-33
def test_chisq_on_reference_fixture(engine, oracle, kw):
    rows, Y, ps = load_fixture(oracle)
    f, fo = flt_pair(oracle, **kw)
    n_out, ids, chi2, pv = (x.cpu().numpy() for x in engine.chisq(to_dev(rows), ps, f))
    for l, (_, _, counts) in enumerate(rows):
        a, rid, rc, rp = oracle.chisq_locus(counts, ps, fo)
        assert n_out[l] == a
        if a:
            assert ids[l, :a].tolist() == rid.tolist()
            assert abs(chi2[l] - rc) <= 1e-10 * max(1.0, abs(rc)) and abs(pv[l] - rp) <= 1e-10


def test_reference_unit_test_vectors_through_the_gpu(engine, oracle):
    # correlation_test.rs:138-181 and chisq_test.rs:57-81 as 1-locus batches
    from poolgen_amd import Filter
    c = np.zeros((1, 5, 6), dtype=np.int32); c[0, :, :2] = [[1, 9], [2, 8], [3, 7], [4, 6], [5, 5]]
    f = Filter(min_allele_frequency=0.005)
    n_out, ids, mf, r, p = (x.cpu().numpy() for x in engine.correlation(torch.from_numpy(c).cuda(), [20.0] * 5, f,
                                                                        np.array([2.0, 1.0, 1.0, 5.0, 2.0])))
    assert n_out[0] == 1 and AL[ids[0, 0]] == "A" and rustfmt.display(mf[0, 0]) == "0.3"
    assert rustfmt.roundup_own(r[0, 0, 0], 6) == "0.3849" and abs(p[0, 0, 0] - 0.5223146158470686) < 1e-12
    c = np.zeros((1, 4, 6), dtype=np.int32); c[0, :, :2] = [[0, 20], [20, 0], [0, 20], [20, 0]]
    n_out, ids, chi2, pv = (x.cpu().numpy() for x in engine.chisq(torch.from_numpy(c).cuda(), [0.2] * 4, f))
    assert n_out[0] == 2 and "".join(AL[i] for i in ids[0, :2]) == "AT"
    assert rustfmt.roundup_own(chi2[0], 6) == "4" and abs(pv[0] - 0.7797774084757156) < 1e-12


@pytest.mark.parametrize("n,L,kw", [(200, 3000, dict()), (33, 2000, dict(maf=0.05)), (100, 2500, dict(remove_ns=False)),
                                    (7, 1500, dict(min_cov=30)), (24, 1500, dict(min_cov=0, miss=0.2)),
                                    (16, 700, dict(min_cov=0, miss=0.5, remove_ns=False))])
def test_synthetic_batches(engine, oracle, n, L, kw, ols_kernel):
    from poolgen_amd import synth
    counts = synth.sync_counts(L, n, "cuda", seed=99)
    # add third alleles, Ns, deletions and a few degenerate loci so that every filter branch fires
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    counts[:, :, 2] = (torch.rand(L, n, generator=g, device="cuda") < 0.15).int() * torch.randint(0, 9, (L, n), generator=g, device="cuda", dtype=torch.int32)
    counts[:, :, 4] = (torch.rand(L, n, generator=g, device="cuda") < 0.05).int()
    counts[::97, :, 5] = 3
    counts[5::211, :, 1] = 0; counts[5::211, :, 2] = 0       # monomorphic -> dropped (< 2 alleles)
    counts[11::307, 0, :] = 0                                # a pool without coverage -> dropped by min depth
    counts[17::131, :min(6, n - 1), :] = 0                   # several uncovered pools: the missingness threshold
    counts[13::401, :, 0] = 10; counts[13::401, :, 1] = 10; counts[13::401, :, 2] = 0   # constant frequency: singular
    Y = synth.phenotypes(synth.genotype_matrix(64, n, "cuda", seed=99), n, k=3, seed=4)
    ps = np.linspace(10, 30, n)
    f, fo = flt_pair(oracle, **kw)
    rows = counts.cpu().numpy().astype(np.uint64)
    check_stat_op(engine.ols_iterate(counts, ps, f, Y), oracle.ols_iterate_locus, rows, Y, ps, fo, oracle=oracle)
    check_stat_op(engine.correlation(counts, ps, f, Y), oracle.correlation_locus, rows, Y, ps, fo,
                  stat_rtol=0, stat_atol=1.0000001e-7)
    n_out, ids, chi2, pv = (x.cpu().numpy() for x in engine.chisq(counts, ps, f))
    for l in range(L):
        a, rid, rc, rp = oracle.chisq_locus(rows[l], ps, fo)
        assert n_out[l] == a
        if a:
            assert ids[l, :min(a, 5)].tolist() == rid.tolist()[:5]
            if np.isnan(rc):   # an uncovered pool (allowed by max_missingness_rate) makes the table NaN on both sides
                assert np.isnan(chi2[l]) and np.isnan(pv[l]) and np.isnan(rp)
                continue
            assert abs(chi2[l] - rc) <= 1e-10 * max(1.0, abs(rc)) and abs(pv[l] - rp) <= 1e-10


@pytest.mark.parametrize("n,L,kpm1,kw", [(5, None, False, dict()), (5, None, True, dict(min_cov=10, maf=0.01)),
                                         (40, 1200, True, dict(maf=0.02)), (33, 900, False, dict(remove_ns=False)),
                                         (24, 700, True, dict(min_cov=0, miss=0.3))])
def test_loader_against_oracle(engine, oracle, n, L, kpm1, kw):
    """pg_load_plan_dev / pg_load_emit_dev = FileSyncPhen::load + into_genotypes_and_phenotypes (sync.rs:972-1180):
    the set of columns, their alleles and every frequency are bit-exact (integer counts, IEEE divisions)."""
    from poolgen_amd import synth
    if L is None:
        rows, _, ps = load_fixture(oracle)
        counts = to_dev(rows)
        host = [r[2] for r in rows]
    else:
        counts = synth.sync_counts(L, n, "cuda", seed=7)
        g = torch.Generator(device="cuda"); g.manual_seed(11)
        counts[:, :, 2] = (torch.rand(L, n, generator=g, device="cuda") < 0.2).int() * torch.randint(0, 7, (L, n), generator=g, device="cuda", dtype=torch.int32)
        counts[:, :, 4] = (torch.rand(L, n, generator=g, device="cuda") < 0.05).int()
        counts[3::101, 0, :] = 0
        counts[9::57, :, 1] = 0; counts[9::57, :, 2] = 0
        ps = np.linspace(10, 30, n)
        host = list(counts.cpu().numpy().astype(np.uint64))
    f, fo = flt_pair(oracle, **kw)
    nloc = len(host)
    rng = np.random.default_rng(3)
    order = rng.permutation(nloc)
    pool_keep = np.ones(n, dtype=bool); pool_keep[1::4] = False
    G, col_locus, col_allele = engine.load_frequencies(counts, ps, f, keep_p_minus_1=kpm1, order=torch.from_numpy(order),
                                                       pool_keep=pool_keep)
    G, col_locus, col_allele = G.cpu().numpy(), col_locus.cpu().numpy(), col_allele.cpu().numpy()
    want_cols, want_loc, want_al = [], [], []
    for l in order:
        res = oracle.filter_locus(host[l], ps, fo)
        if res is None:
            continue
        ids, fc = res
        fr = oracle.to_frequencies(fc)
        if kpm1:
            fr, ids = oracle.sort_by_allele_freq(fr, ids, True)
            fr, ids = fr[:, 1:], ids[1:]
        for j, a in enumerate(ids):
            want_cols.append(fr[pool_keep, j]); want_loc.append(int(l)); want_al.append(int(a))
    assert len(want_cols) == G.shape[0] > 0
    assert col_locus.tolist() == want_loc and col_allele.tolist() == want_al
    nk = int(pool_keep.sum())
    assert np.array_equal(G[:, :nk], np.stack(want_cols), equal_nan=True)
    assert np.all(G[:, nk:] == 0.0)


def test_loader_first_locus_reference_literal(engine, oracle):
    """base/sync.rs:1516-1535, :1616: the first locus of tests/test.sync as the reference's own loader test sees it."""
    import json
    g = json.loads((GOLD / "reference_literals.json").read_text())["loaded_first_locus"]
    rows, _, ps = load_fixture(oracle)
    from poolgen_amd import Filter
    G, col_locus, col_allele = engine.load_frequencies(to_dev(rows), ps, Filter(min_allele_frequency=0.005), keep_p_minus_1=True)
    first = (col_locus == 0).nonzero().flatten().tolist()
    assert (rows[0][0], rows[0][1]) == (g["chromosome"], g["position"])
    assert "".join(AL[int(col_allele[c])] for c in first) == g["alleles"]
    assert G[first, :5].T.reshape(-1).tolist() == g["freq"]


def test_config2_full_size_properties(engine, oracle, ols_kernel):
    """BASELINE config 2 shape (100 pools x 1 M loci from counts): size-independent properties of the three operators.
    A locus' result depends on that locus only: processing the batch in another order permutes the results bit for bit
    (all alignment classes, units, second-pass lists and compact records differ between the two runs); relabelling the
    pools together with their sizes and phenotypes leaves chi-square untouched up to summation order; a sampled slice
    agrees with the oracle."""
    from poolgen_amd import synth
    n, L = 100, 1_000_000
    counts = synth.sync_counts(L, n, "cuda", seed=123)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    third = (torch.rand(L, 1, generator=g, device="cuda") < 0.1).int()
    counts[:, :, 2] = third * torch.randint(0, 9, (L, n), generator=g, device="cuda", dtype=torch.int32)
    Y = synth.phenotypes(synth.genotype_matrix(64, n, "cuda", seed=7), n, k=1, seed=2)
    ps = np.full(n, 20.0)
    f, fo = flt_pair(oracle)
    perm = torch.randperm(L, generator=g, device="cuda")
    shuffled = counts[perm].contiguous()
    def same(x, y):
        return (x == y) | (torch.isnan(x) & torch.isnan(y)) if x.is_floating_point() else (x == y)

    for run in (lambda c: engine.ols_iterate(c, ps, f, Y), lambda c: engine.correlation(c, ps, f, Y), lambda c: engine.chisq(c, ps, f)):
        a = [x[perm] for x in run(counts)]
        b = run(shuffled)
        n_out = b[0]
        assert bool((a[0] == n_out).all()) and int((n_out > 0).sum()) > 0.9 * L
        for x, y in zip(a[1:], b[1:]):
            if x.dim() == 1:                                   # chi-square statistic / p-value: one per emitted locus
                used = n_out > 0
            else:                                              # per emitted allele (x traits): entries beyond n_out are unspecified
                used = torch.arange(x.shape[1], device="cuda")[None, :] < n_out[:, None]
                if x.dim() == 3:
                    used = used[:, :, None].expand_as(x)
            assert bool(same(x, y)[used].all())
    # chi-square under a relabelling of the pools
    pp = torch.randperm(n, generator=g, device="cuda")
    n1, i1, c1, p1 = engine.chisq(counts[:200_000].contiguous(), ps, f)
    n2, i2, c2, p2 = engine.chisq(counts[:200_000][:, pp].contiguous(), ps, f)
    assert bool((n1 == n2).all()) and torch.allclose(c1, c2, rtol=1e-10, atol=1e-12, equal_nan=True)
    # a slice against the oracle
    sl = slice(500_000, 500_000 + 512)
    rows = counts[sl].cpu().numpy().astype(np.uint64)
    res = engine.ols_iterate(counts, ps, f, Y)
    check_stat_op(tuple(x[sl] for x in res), oracle.ols_iterate_locus, rows, Y, ps, fo, oracle=oracle)


def test_slot_major_layout_of_the_abi(engine, oracle):
    """include/poolgen_hip.h: element (slot r, locus l) of allele_ids / mean_freq at r * L + l, of stat / pval at
    (r * L + l) * k + trait; only the slots r < n_out[l] are specified.  The raw arrays of the library against the
    locus-major copies the other tests index, on the reference's fixture (multi-allelic loci included) with two traits."""
    rows, Y, ps = load_fixture(oracle)
    f, _ = flt_pair(oracle)
    Y2 = np.column_stack([Y[:, 0], Y[:, 0] ** 2 + 0.1 * np.arange(Y.shape[0])])
    Y3 = np.column_stack([Y2, np.cos(np.arange(Y.shape[0]))])   # three traits: ols_iter takes them as two launches (2 + 1)
    counts = to_dev(rows)
    L = counts.shape[0]
    for op, Yk in ((engine.ols_iterate, Y2), (engine.correlation, Y2), (engine.ols_iterate, Y3)):
        k = Yk.shape[1]
        n_out, ids, mf, stat, pv = (x.cpu().numpy() for x in op(counts, ps, f, Yk))
        rn, rids, rmf, rstat, rpv = (x.cpu().numpy() for x in op(counts, ps, f, Yk, raw=True))
        assert rids.shape == (5, L) and rmf.shape == (5, L) and rstat.shape == (5, L, k) and rpv.shape == (5, L, k)
        if k == 3:   # the third trait's column equals a one-trait call on it (the launch groups write disjoint trait columns)
            s1 = op(counts, ps, f, Y3[:, 2:3])[3].cpu().numpy()
            assert np.array_equal(stat[:, :, 2], s1[:, :, 0], equal_nan=True)
        assert np.array_equal(rn, n_out) and n_out.max() >= 2   # the fixture has loci that emit several rows
        for r in range(5):
            live = n_out > r
            assert np.array_equal(rids[r][live], ids[live, r])
            assert np.array_equal(rmf[r][live], mf[live, r], equal_nan=True)
            assert np.array_equal(rstat[r][live], stat[live, r], equal_nan=True)
            assert np.array_equal(rpv[r][live], pv[live, r], equal_nan=True)
            assert np.all(ids[~live, r] == -1) and np.all(np.isnan(mf[~live, r]))   # the copies fill what the library leaves open
    n_out, ids, chi2, pv = (x.cpu().numpy() for x in engine.chisq(counts, ps, f))
    rn, rids, rchi2, rpv = (x.cpu().numpy() for x in engine.chisq(counts, ps, f, raw=True))
    assert rids.shape == (5, L) and np.array_equal(rn, n_out)
    for r in range(5):
        live = n_out > r
        assert np.array_equal(rids[r][live], ids[live, r])


@pytest.mark.parametrize("n,L,err,kw", [(100, 3000, 0.005, dict(maf=0.01)), (100, 2000, 0.001, dict()), (200, 1500, 0.005, dict()),
                                        (33, 2001, 0.01, dict(maf=0.02)), (48, 1500, 0.02, dict(maf=0.05, remove_ns=False)),
                                        (16, 1200, 0.01, dict(maf=0.01, min_cov=0, miss=0.3)), (7, 900, 0.01, dict(maf=0.01))])
def test_error_bearing_counts(engine, oracle, n, L, err, kw, ols_kernel):
    """Realistic counts: every read misread with probability `err` onto one of the other five sync columns, so nearly every
    locus carries reads of alleles the MAF filter drops.  The reference filters and THEN recomputes the frequencies on the
    filtered counts (gwas/ols.rs:210-230 -> base/sync.rs:252-286, :166-192): one dropped read changes every denominator of
    its pool.  Emission and allele ids bit-exact, mean frequencies bit-exact, statistics 1e-10 -- for the loci the streaming
    pass closes from its speculated pair, for the mis-speculated ones and for the multi-allelic ones (error alleles that
    survive the filter: the (200, default maf) and 7-pool cases) alike."""
    from poolgen_amd import synth
    counts = synth.sync_counts(L, n, "cuda", seed=41, error_rate=err)
    g = torch.Generator(device="cuda"); g.manual_seed(17)
    counts[3::89, : max(1, n // 5), :] = 0                        # uncovered pools (missingness; NaN frequencies where allowed)
    counts[7::113, :, 1] = 0                                       # the minor allele gone: only error alleles beside the major one
    hole = (torch.rand(L, n, generator=g, device="cuda") < 0.3 / n)  # pools where only stray reads remain: uncovered over the survivors
    counts[:, :, 0] *= (~hole).int(); counts[:, :, 1] *= (~hole).int()
    Y = synth.phenotypes(synth.genotype_matrix(64, n, "cuda", seed=41), n, k=2, seed=6)
    ps = np.linspace(10, 30, n)
    f, fo = flt_pair(oracle, **kw)
    rows = counts.cpu().numpy().astype(np.uint64)
    check_stat_op(engine.ols_iterate(counts, ps, f, Y), oracle.ols_iterate_locus, rows, Y, ps, fo, oracle=oracle)
    check_stat_op(engine.correlation(counts, ps, f, Y), oracle.correlation_locus, rows, Y, ps, fo,
                  stat_rtol=0, stat_atol=1.0000001e-7)
    n_out, ids, chi2, pv = (x.cpu().numpy() for x in engine.chisq(counts, ps, f))
    emitted = 0
    for l in range(L):
        a, rid, rc, rp = oracle.chisq_locus(rows[l], ps, fo)
        assert n_out[l] == a, f"chisq locus {l}"
        if a:
            emitted += 1
            assert ids[l, :min(a, 5)].tolist() == rid.tolist()[:5]
            if np.isnan(rc):
                assert np.isnan(chi2[l]) and np.isnan(pv[l]) and np.isnan(rp)
                continue
            assert abs(chi2[l] - rc) <= 1e-10 * max(1.0, abs(rc)) and abs(pv[l] - rp) <= 1e-10, f"chisq locus {l}"
    assert emitted > L // 4


def test_poisoned_tail_behind_an_odd_batch(engine, oracle):
    """with L * n odd the last 16-byte load of the streaming pass reaches 8 bytes past the batch; whatever
    lies there becomes counts of a locus that does not exist and must neither reach a result nor raise the 2^29 complaint."""
    from poolgen_amd import synth
    n, L = 33, 2001
    big = torch.full(((L + 40) * n * 6,), -1, dtype=torch.int32, device="cuda")      # 0xFFFFFFFF words behind the batch
    counts = big[: L * n * 6].view(L, n, 6)
    counts.copy_(synth.sync_counts(L, n, "cuda", seed=5, error_rate=0.002))
    assert counts.data_ptr() % 16 == 0 and (L * n * 24) % 16 == 8
    Y = synth.phenotypes(synth.genotype_matrix(64, n, "cuda", seed=5), n, k=1, seed=3)
    ps = np.full(n, 20.0)
    f, fo = flt_pair(oracle)
    rows = counts.cpu().numpy().astype(np.uint64)
    check_stat_op(engine.ols_iterate(counts, ps, f, Y), oracle.ols_iterate_locus, rows, Y, ps, fo, oracle=oracle)
    n_out = engine.chisq(counts, ps, f)[0]
    assert int((n_out > 0).sum()) > L // 2
    G, col_locus, col_allele = engine.load_frequencies(counts, ps, f)
    assert G.shape[0] >= L


@pytest.mark.parametrize("n", [40, 100])
def test_filter_decision_on_the_threshold(engine, oracle, n, ols_kernel):
    """The streaming pass decides q < maf from a cheaper evaluation of q (fma(c, w / rs, q): within (n + 8) ulp of the reference's
    sequential sum of fl(c / rs) * w, base/sync.rs:258-271) and recomputes q literally for loci within 8 (n + 16) eps of a
    threshold.  Thresholds placed ON a locus' own q (to the bit), one ulp above and one below: the decisions must be the oracle's."""
    from poolgen_amd import Filter, synth
    L = 600
    counts = synth.sync_counts(L, n, "cuda", seed=77, error_rate=0.004)
    Y = synth.phenotypes(synth.genotype_matrix(64, n, "cuda", seed=77), n, k=1, seed=1)
    ps = np.linspace(10, 30, n)
    w = ps / ps.sum()
    c = counts.cpu().numpy().astype(np.float64)[:, :, [0, 1, 2, 3, 5]]
    rs = c.sum(axis=2)
    q = np.zeros((L, 5))
    for i in range(n):                        # the reference's order: pools sequentially, multiply then add
        q = q + (c[:, i, :] / rs[:, i, None]) * w[i]
    rows = counts.cpu().numpy().astype(np.uint64)
    tested = 0
    for l, j in ((3, 1), (17, 2), (101, 0), (333, 5 - 1)):
        for maf in (q[l, j], np.nextafter(q[l, j], 1.0), np.nextafter(q[l, j], 0.0), 1.0 - q[l, j]):
            if not (0.0 < maf < 0.5):
                continue
            f, fo = Filter(min_allele_frequency=float(maf)), oracle.filt(True, 1, float(maf), 0.0)
            check_stat_op(engine.ols_iterate(counts, ps, f, Y), oracle.ols_iterate_locus, rows, Y, ps, fo, oracle=oracle)
            n_out, ids, chi2, pv = (x.cpu().numpy() for x in engine.chisq(counts, ps, f))
            for ll in (l, max(l - 1, 0), min(l + 1, L - 1)):
                a, rid, rc, rp = oracle.chisq_locus(rows[ll], ps, fo)
                assert n_out[ll] == a and (a == 0 or ids[ll, :a].tolist() == rid.tolist())
            tested += 1
    assert tested >= 8


def test_second_pass_routes_agree(engine, oracle, monkeypatch, ols_kernel):
    """The second pass takes a short list as it is (tiles of mixed survivor counts) and groups a long one by the number of
    survivors first (k_locus_hist / k_locus_sort): both routes forced on one batch with 2 .. 5 survivors per locus must give the
    same bits (the arithmetic of a locus does not depend on the code variant its tile runs), and the oracle's answers."""
    from poolgen_amd import synth
    n, L = 100, 6000
    counts = synth.sync_counts(L, n, "cuda", seed=13, error_rate=0.005)     # default maf: half of the error alleles survive
    Y = synth.phenotypes(synth.genotype_matrix(64, n, "cuda", seed=13), n, k=2, seed=2)
    ps = np.full(n, 20.0)
    f, fo = flt_pair(oracle)
    rows = counts.cpu().numpy().astype(np.uint64)
    outs = {}
    for route in ("0", "1"):
        monkeypatch.setenv("POOLGEN_LOCUS_GROUPED", route)
        outs[route] = [tuple(x.clone() for x in engine.chisq(counts, ps, f, raw=True)),
                       tuple(x.clone() for x in engine.ols_iterate(counts, ps, f, Y, raw=True)),
                       tuple(x.clone() for x in engine.correlation(counts, ps, f, Y, raw=True))]
        loci, listed = engine.last_listed()
        assert loci == L and listed > L // 2      # (pearson_corr: every locus with three or more survivors goes to the second pass)
    for a, b in zip(outs["0"], outs["1"]):
        n_out = a[0]
        assert torch.equal(n_out, b[0])
        for x, y in zip(a[1:], b[1:]):
            if x.dim() == 1:
                live = n_out > 0
            else:
                live = torch.arange(x.shape[0], device="cuda")[:, None] < n_out[None, :]
                if x.dim() == 3:
                    live = live[:, :, None].expand_as(x)
            same = (x == y) | (torch.isnan(x) & torch.isnan(y)) if x.is_floating_point() else (x == y)
            assert bool(same[live].all())
    monkeypatch.setenv("POOLGEN_LOCUS_GROUPED", "1")
    check_stat_op(engine.ols_iterate(counts, ps, f, Y), oracle.ols_iterate_locus, rows[:1500], Y, ps, fo, oracle=oracle)


def test_kernel_choice_follows_the_data(engine, oracle, monkeypatch):
    """Left alone (no POOLGEN_OLS_ITER_KERNEL) a context starts with the order-free kernel, moves to the streaming pass after a clean
    batch and back after an error-bearing one; whatever ran, emission and ids are the oracle's and the numbers within 1e-10."""
    from poolgen_amd import synth
    monkeypatch.delenv("POOLGEN_OLS_ITER_KERNEL", raising=False)
    n, L = 100, 4000
    Y = synth.phenotypes(synth.genotype_matrix(64, n, "cuda", seed=3), n, k=1, seed=3)
    ps = np.full(n, 20.0)
    f, fo = flt_pair(oracle, maf=0.01)
    clean = synth.sync_counts(L, n, "cuda", seed=21)
    dirty = synth.sync_counts(L, n, "cuda", seed=21, error_rate=0.005)
    for counts in (clean, clean, dirty, dirty, clean):
        res = engine.ols_iterate(counts, ps, f, Y)
        check_stat_op(tuple(x[:600] for x in res), oracle.ols_iterate_locus, counts[:600].cpu().numpy().astype(np.uint64), Y, ps, fo, oracle=oracle)


@pytest.mark.parametrize("n", [100, 200, 37])
def test_chisq_register_and_buffer_variants_agree(engine, oracle, n, monkeypatch):
    """chisq_test's order-free pass reads its pools straight into registers (no LDS staging buffer); POOLGEN_ROWS_DIRECT=0 keeps the
    buffered variant ols_iter uses.  Same pool-to-lane assignment, same butterflies: the two must give the same bits."""
    from poolgen_amd import synth
    L = 20011
    counts = synth.sync_counts(L, n, "cuda", seed=29, error_rate=0.004)
    counts[11::97, : max(1, n // 7), :] = 0
    ps = np.linspace(12, 40, n)
    f, fo = flt_pair(oracle, maf=0.002)
    monkeypatch.setenv("POOLGEN_OLS_ITER_KERNEL", "rows")
    a = [x.cpu().numpy() for x in engine.chisq(counts, ps, f)]
    monkeypatch.setenv("POOLGEN_ROWS_DIRECT", "0")
    b = [x.cpu().numpy() for x in engine.chisq(counts, ps, f)]
    assert int((a[0] > 0).sum()) > L // 2
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    rows = counts[:300].cpu().numpy().astype(np.uint64)
    for l in range(300):
        k, rid, rc, rp = oracle.chisq_locus(rows[l], ps, fo)
        assert a[0][l] == k
        if k and not np.isnan(rc):
            assert abs(a[2][l] - rc) <= 1e-10 * max(1.0, abs(rc)) and abs(a[3][l] - rp) <= 1e-10
