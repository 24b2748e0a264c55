"""End-to-end: the C++ `poolgen` CLI (hot subcommands) on the reference's fixture (BASELINE
config 1, the CI invocations of .github/workflows/rust.yml:32-37) against CSV text produced by
the oracle.  Text fields must be identical; floats are compared numerically where the reference
prints full precision (1e-16-level differences of the p-value algorithm are not printable-stable)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CLI = ROOT / "poolgen_amd" / "csrc" / "poolgen"
GOLD = Path(__file__).parent / "golden"
PS = [20.0] * 5


def run_cli(*args, ok=True):
    r = subprocess.run([str(CLI), *map(str, args)], capture_output=True, text=True)
    assert (r.returncode == 0) == ok, r.stderr
    return r


def fixture(oracle):
    rows = []
    for line in (GOLD / "test.sync").read_text().splitlines():
        n, chrom, pos, counts = oracle.parse_sync_line(line)
        if n > 0:
            rows.append((chrom, pos, counts))
    Y = np.loadtxt(GOLD / "test.csv", delimiter=",", comments="#", usecols=(2, 3))
    return rows, Y


def compare_csv(got, want, float_cols, max_noise_rows, exempt=None):
    """exempt: set of (chromosome, position) whose rows may differ numerically (their text fields must still agree): the
    loci the caller has shown to be rank-deficient, where the reference prints rounding noise.  Every other row must agree."""
    g, w = got.splitlines(), want.splitlines()
    assert len(g) == len(w), (len(g), len(w))
    assert g[0] == w[0]
    noisy = 0
    for a, b in zip(g[1:], w[1:]):
        if a == b:
            continue
        fa, fb = a.split(","), b.split(",")
        assert len(fa) == len(fb)
        same = True
        for i, (x, y) in enumerate(zip(fa, fb)):
            if i in float_cols:
                xv, yv = float(x), float(y)
                # a column printed on a d-decimal grid can differ by one unit of that grid when the
                # value sits on a rounding boundary (e.g. chi2 = 2.4609375 -> 2.460937 | 2.460938)
                if not ((np.isnan(xv) and np.isnan(yv)) or abs(xv - yv) <= float_cols[i] * max(1.0, abs(yv))):
                    same = False
            else:
                assert x == y, (a, b)
        if not same and exempt is not None and (fa[0], int(fa[1])) not in exempt:
            raise AssertionError(f"row of a full-rank locus differs: {a} | {b}")
        noisy += 0 if same else 1
        if not same:
            print("DIFF", a, "|", b)
    assert noisy <= max_noise_rows, f"{noisy} rows beyond 1e-10"
    return noisy


def rank_deficient_loci(oracle, rows, f):
    """(chromosome, position) of the loci whose ols_iter design [1 | sorted frequencies without the major allele] is numerically
    rank deficient (cond > 1e7, i.e. cond(X'X) > 1e14: duplicated pools / alleles at n = 5): there, and only there, the
    reference's LU prints rounding noise and only the emission pattern can be compared."""
    bad = set()
    for c, p, cnt in rows:
        res = oracle.filter_locus(cnt, PS, f)
        if res is None:
            continue
        ids, fc = res
        fr, ids = oracle.sort_by_allele_freq(oracle.to_frequencies(fc), ids, True)
        X = np.ones_like(fr); X[:, 1:] = fr[:, 1:]
        if not np.all(np.isfinite(X)) or np.linalg.cond(X) > 1e7:
            bad.add((c, int(p)))
    return bad


@pytest.mark.parametrize("extra,pieces", [([], None), (["--min-coverage-depth", "10", "--min-allele-frequency", "0.01"], None),
                                          ([], "30000")])
def test_cli_ols_iter(oracle, tmp_path, extra, pieces):
    import os
    rows, Y = fixture(oracle)
    out = tmp_path / "o.csv"
    env = dict(os.environ, PGH_STREAM_CHUNK_BYTES=pieces) if pieces else None   # small pieces: boundaries under the same zero budget
    r = subprocess.run([str(CLI), "ols_iter", "-f", str(GOLD / "test.sync"), "-p", str(GOLD / "test.csv"), "--phen-delim", ",",
                        "--phen-name-col", "0", "--phen-value-col", "2,3", "--n-threads", "2", "-o", str(out), *extra],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().endswith(str(out))
    f = oracle.filt(True, 10, 0.01, 0.0) if extra else oracle.filt()
    want = "#chr,pos,alleles,freq,phenotype,statistic,pvalue\n" + "".join(
        oracle.ols_iterate_csv(c, p, cnt, Y, PS, f) or "" for c, p, cnt in rows)
    # rows may differ numerically ONLY at loci the oracle's own design matrix shows to be rank deficient; everywhere else: zero
    exempt = rank_deficient_loci(oracle, rows, f)
    noisy = compare_csv(out.read_text(), want, {3: 1.0000001e-8, 5: 1.0000001e-6, 6: 1e-10}, max_noise_rows=len(want.splitlines()),
                        exempt=exempt)
    print(f"ols_iter CSV: {noisy} rows differ numerically, all at the {len(exempt)} rank-deficient loci of {len(rows)}")
    assert len(exempt) < len(rows) // 20
    run_cli("ols_iter", "-f", GOLD / "test.sync", "-p", GOLD / "test.csv", "--phen-value-col", "2,3", "-o", out, ok=False)  # create_new


def test_cli_pearson_and_chisq(oracle, tmp_path):
    rows, Y = fixture(oracle)
    f = oracle.filt()
    out = tmp_path / "p.csv"
    run_cli("pearson_corr", "-f", GOLD / "test.sync", "-p", GOLD / "test.csv", "--phen-value-col", "2,3", "--n-threads", 3, "-o", out)
    want = "#chr,pos,alleles,freq,phenotype,statistic,pvalue\n" + "".join(
        oracle.correlation_csv(c, p, cnt, Y, PS, f) or "" for c, p, cnt in rows)
    g = out.read_text()
    assert g.count("\n") == want.count("\n")
    for a, b in zip(g.splitlines()[1:], want.splitlines()[1:]):
        fa, fb = a.split(","), b.split(",")
        assert fa[:3] == fb[:3] and fa[4] == fb[4] and fa[3] == fb[3]          # labels and mean frequency text
        assert abs(float(fa[5]) - float(fb[5])) <= 1.0000001e-6                # r printed at 6 dp
        pa, pb = float(fa[6]), float(fb[6])
        assert (np.isnan(pa) and np.isnan(pb)) or abs(pa - pb) <= 1e-10
    out = tmp_path / "c.csv"
    run_cli("chisq_test", "-f", GOLD / "test.sync", "-p", GOLD / "test.csv", "--n-threads", 2, "-o", out)
    want = "#chr,pos,alleles,statistic,pvalue\n" + "".join(oracle.chisq_csv(c, p, cnt, PS, f) or "" for c, p, cnt in rows)
    compare_csv(out.read_text(), want, {3: 1.0000001e-6, 4: 1e-10}, max_noise_rows=0)


@pytest.mark.parametrize("thr,keep1", [(0.75, False), (0.75, True), (0.8, True)])
def test_cli_ols_iter_with_kinship(oracle, exact, tmp_path, thr, keep1):
    rows, Y = fixture(oracle)
    f = oracle.filt()
    out = tmp_path / "k.csv"
    args = ["ols_iter_with_kinship", "-f", GOLD / "test.sync", "-p", GOLD / "test.csv", "--phen-value-col", "2,3",
            "--n-threads", 2, "-x", thr, "-o", out]
    if keep1:
        args.append("--keep-p-minus-1")
    run_cli(*args)
    # loader (sync.rs:1044-1179): filter, frequencies, sort loci by (chromosome, position)
    lab, cols = [("intercept", 0, "intercept")], []
    for chrom, pos, cnt in sorted(rows, key=lambda r: (r[0], r[1])):
        res = oracle.filter_locus(cnt, PS, f)
        if res is None:
            continue
        ids, fc = res
        fr = oracle.to_frequencies(fc)
        if keep1:
            fr, ids = oracle.sort_by_allele_freq(fr, ids, True)
            fr, ids = fr[:, 1:], ids[1:]
        for j, a in enumerate(ids):
            lab.append((chrom, pos, "ATCGND"[a])); cols.append(fr[:, j])
    G = np.array(cols)
    ref = oracle.ols_with_covariate(G, Y, thr)
    lines = out.read_text().splitlines()
    assert lines[0] == "#chr,pos,alleles,phenotype,statistic,pvalue" and len(lines) == 1 + 2 * len(cols)
    assert lines[1].startswith("intercept,0,intercept,Pheno_0,")                  # label shift (gwas/ols.rs:421-425)
    # m = 0: 1e-10.  m >= 1 with FIVE pools (P = m + 2 >= 3 columns, <= 2 residual degrees of freedom, covariates that are eigenvectors
    # of a 5 x 5 kinship known to 1e-16): no fp64 chain reaches 1e-10 on every column -- the product is 1.1e-10 .. 4e-10 from
    # binary128 on a handful of them, the literal oracle 1e-8; asserted at 1e-9 here, at 1e-10 where n is realistic (test_gpu_exact.py)
    tol = 1e-10 if ref["m"] == 0 else 1e-9
    # The numbers are checked against the same chain in binary128 (tests/test_gpu_exact.py: with 5 pools many columns are nearly
    # constant and the literal normal equations of the oracle lose cond * eps digits there -- 2e-10 on a coefficient of 59 -- as
    # they do on the covariate fits), the p-values against the reference's formula at the binary128 t; the oracle contributes the
    # pattern of failed fits (its LU hit an exact zero).
    from test_gpu_exact import formula_p
    ex = exact.ols_with_covariate(G, Y, thr)
    assert ex["m"] == ref["m"]
    nanpat = np.isnan(ref["beta"])
    ref = dict(m=ex["m"], beta=np.where(nanpat, np.nan, ex["beta"]), pval=np.where(nanpat, np.nan, formula_p(oracle, ex, G.shape[1])))
    # Rows may differ ONLY for columns that are (numerically) constant over the 5 pools: there the reference's LU happens to hit an
    # exact zero pivot (NaN) or not (rounding noise) depending on the residue, while the product flags s_gg <= 1e-12 g'g as NaN
    # always.  The exemption is computed from the column itself; every other row must agree.
    gc = G - G.mean(axis=1, keepdims=True)
    flat = (gc * gc).sum(axis=1) <= 1e-10 * (G * G).sum(axis=1)
    bad = 0
    for j in range(2):
        for i in range(len(cols)):
            fa = lines[1 + j * len(cols) + i].split(",")
            assert (fa[0], int(fa[1]), fa[2], fa[3]) == (lab[i][0], lab[i][1], lab[i][2], f"Pheno_{j}")
            b, p = float(fa[4]), float(fa[5])
            rb, rp = ref["beta"][i, j], ref["pval"][i, j]
            okb = (np.isnan(b) and np.isnan(rb)) or abs(b - rb) <= tol + tol * abs(rb)   # rtol 1e-10 + atol 1e-10, as everywhere
            okp = (np.isnan(p) and np.isnan(rp)) or abs(p - rp) <= tol
            if not (okb and okp):
                assert flat[i], f"row of a non-constant column differs: {lines[1 + j * len(cols) + i]} | oracle {rb} {rp}"
                assert np.isnan(b) and np.isnan(p)          # ... and on a constant column the product's answer is NaN, never noise
                bad += 1
    print(f"kinship CSV: {bad} rows NaN here / noise in the reference, all on the {int(flat.sum())} constant columns of {len(cols)}")
    assert flat.sum() < len(cols) // 10


def test_cli_default_output_name_and_errors(tmp_path):
    import shutil
    shutil.copy(GOLD / "test.sync", tmp_path / "my.data.sync")
    r = run_cli("chisq_test", "-f", tmp_path / "my.data.sync", "-p", GOLD / "test.csv")
    name = r.stdout.strip().splitlines()[-1]
    assert name.startswith(str(tmp_path / "my.data-")) and name.endswith("-chisq_test.csv") and Path(name).exists()
    assert run_cli("ridge_iter", "-f", GOLD / "test.sync", "-p", GOLD / "test.csv", ok=False).stderr.count("Invalid analysis")
    assert run_cli("chisq_test", "-f", GOLD / "test.sync", "-p", GOLD / "test.csv", "--min-allele-frequency", "1.5", ok=False).returncode == 1
    # flag values are read like Rust's parse::<usize / f64>(): no trailing garbage, no negative unsigned, no hex floats
    for flag, val in (("--n-threads", "2x"), ("--min-coverage-depth", "-1"), ("--min-allele-frequency", "0x0.1"), ("--phen-value-col", "2,three"),
                      ("--n-threads", "0")):
        r = run_cli("chisq_test", "-f", GOLD / "test.sync", "-p", GOLD / "test.csv", flag, val, "-o", tmp_path / "bad.csv", ok=False)
        assert r.returncode == 1 and val.split(",")[-1] in r.stderr and not (tmp_path / "bad.csv").exists(), (flag, val, r.stderr)


def test_pileup_input_equals_pileup2sync_then_analysis(tmp_path):
    """A *.pileup input is converted in memory; the result must be byte-identical to running pileup2sync first and
    the analysis on the sync file it writes (the reader's column relabelling, sync.rs:134 vs pileup.rs:184, included)."""
    import random, subprocess, sys
    sys.path.insert(0, str(Path(__file__).parent))
    from test_pileup import _random_line
    rng = random.Random(5)
    n = 5
    lines = [_random_line(rng, n, False) for _ in range(3000)]
    pile = tmp_path / "in.pileup"; pile.write_text("\n".join(lines) + "\n", encoding="latin-1")
    phen = GOLD / "test.csv"
    exe = ROOT / "poolgen_amd" / "csrc" / "poolgen"
    sync = tmp_path / "conv.sync"
    subprocess.run([str(exe), "pileup2sync", "-f", str(pile), "-p", str(phen), "-o", str(sync), "--n-threads", "3"], check=True,
                   capture_output=True)
    assert sync.read_text().count("\n") > 300
    # --keep-ns included: pileup2sync writes its six counts in the order A:T:C:G:D:N (pileup.rs:184) and the sync reader labels
    # columns A,T,C,G,N,D (sync.rs:134), so on a converted file "remove Ns" drops the DELETION counts and keeps the Ns -- the
    # reference's two-step behaviour, which the in-memory path must reproduce with and without the flag
    for analysis, extra in (("ols_iter", ["--phen-value-col", "2,3"]), ("chisq_test", []), ("chisq_test", ["--keep-ns"]),
                            ("ols_iter", ["--phen-value-col", "2", "--keep-ns"]),
                            ("ols_iter_with_kinship", ["--phen-value-col", "2", "-x", "0.5"])):
        tag = analysis + ("_ns" if "--keep-ns" in extra else "")
        a, b = tmp_path / f"{tag}_sync.csv", tmp_path / f"{tag}_pileup.csv"
        src_sync = sync
        if "--keep-ns" in extra:   # the in-memory path hands the flag to the conversion too, as two commands with the same flags would
            src_sync = tmp_path / "conv_ns.sync"
            if not src_sync.exists():
                subprocess.run([str(exe), "pileup2sync", "-f", str(pile), "-p", str(phen), "-o", str(src_sync), "--n-threads", "3", "--keep-ns"],
                               check=True, capture_output=True)
        for src, dst in ((src_sync, a), (pile, b)):
            r = subprocess.run([str(exe), analysis, "-f", str(src), "-p", str(phen), "-o", str(dst), "--n-threads", "2", *extra],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
        ta, tb = a.read_text(), b.read_text()
        if ta != tb:
            la, lb = ta.splitlines(), tb.splitlines()
            first = next(i for i in range(min(len(la), len(lb))) if la[i] != lb[i])
            raise AssertionError(f"{tag}: sync-then-analysis {len(la)} lines vs in-memory {len(lb)}; first difference at line {first}: "
                                 f"[sync] {la[first]} | [pileup] {lb[first]}")
        assert ta.count("\n") > 100, tag


def _read_kinship_csv(path):
    rows = [l.rstrip("\n").split(",") for l in Path(path).read_text().splitlines()[1:]]
    return [r[:4] for r in rows], np.array([[float(r[4]), float(r[5])] for r in rows])


def test_streamed_kinship_equals_whole_file(tmp_path):
    """ols_iter_with_kinship on an input taken in pieces (double-buffered pinned parse, per-piece loader and partial
    kinship): same labels and, up to the summation order of the kinship partials, the same numbers as the whole-file
    path; an unsorted input is refused (the (chromosome, position) sort cannot be done across pieces)."""
    import os, subprocess
    exe = ROOT / "poolgen_amd" / "csrc" / "poolgen"
    lines = [l for l in (GOLD / "test.sync").read_text().splitlines() if not l.startswith("#")]
    lines.sort(key=lambda l: (l.split("\t")[0].encode(), int(l.split("\t")[1])))
    srt = tmp_path / "sorted.sync"; srt.write_text("\n".join(lines) + "\n")
    base = [str(exe), "ols_iter_with_kinship", "-p", str(GOLD / "test.csv"), "--phen-value-col", "2,3", "--n-threads", "3", "-x", "0.5"]
    whole, piece = tmp_path / "whole.csv", tmp_path / "pieces.csv"
    subprocess.run(base + ["-f", str(srt), "-o", str(whole), "--stream-chunk-mb", "0"], check=True, capture_output=True)
    env = dict(os.environ, PGH_STREAM_CHUNK_BYTES="40000")           # ~12 pieces
    r = subprocess.run(base + ["-f", str(srt), "-o", str(piece)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    la, va = _read_kinship_csv(whole)
    lb, vb = _read_kinship_csv(piece)
    assert la == lb and len(la) > 10000
    assert np.allclose(va, vb, rtol=1e-9, atol=1e-12, equal_nan=True)
    r = subprocess.run(base + ["-f", str(GOLD / "test.sync"), "-o", str(tmp_path / "x.csv")], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "sorted by chromosome and position" in r.stderr


@pytest.mark.parametrize("analysis,extra", [("ols_iter", ["--phen-value-col", "2,3"]), ("pearson_corr", ["--phen-value-col", "2,3"]),
                                            ("chisq_test", ["--min-coverage-depth", "5"])])
def test_batch_operators_in_pieces_equal_one_piece(tmp_path, analysis, extra):
    """The per-locus operators always take the input in pieces (parse of piece c + 1 overlaps the GPU and the writer on
    piece c); pieces of 20 KB (some 300 of them for the fixture) must give the same file, byte for byte, as one piece."""
    import os
    base = [str(CLI), analysis, "-f", str(GOLD / "test.sync"), "-p", str(GOLD / "test.csv"), "--n-threads", "3", *extra]
    one, many = tmp_path / "one.csv", tmp_path / "many.csv"
    r = subprocess.run(base + ["-o", str(one)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(base + ["-o", str(many)], capture_output=True, text=True, env={**os.environ, "PGH_STREAM_CHUNK_BYTES": "20000"})
    assert r.returncode == 0, r.stderr
    assert one.read_bytes() == many.read_bytes() and one.stat().st_size > 10000


def test_pieces_whose_text_leaves_the_mapping(tmp_path):
    """Round 4: the text of a parsed piece is dropped from the file mapping on a helper thread (madvise, pieces of 1 MiB or more:
    unmapping 27 GB at exit cost more than the CSV writer).  A 9 MB sorted sync in 2 MiB pieces through ols_iter_with_kinship
    and ols_iter: byte-identical files with the pages dropped, with PGH_KEEP_MAPPED=1, and (ols_iter) in one piece."""
    import os
    from poolgen_amd import synth
    n, L = 120, 6000
    counts = synth.sync_counts(L, n, "cpu", seed=73, error_rate=0.002).numpy()
    sync = tmp_path / "big.sync"
    with open(sync, "w") as fh:
        for l in range(L):
            fh.write(f"chr{1 + l // 2500}\t{10 + 3 * l}\tN\t" + "\t".join(":".join(str(int(x)) for x in counts[l, i]) for i in range(n)) + "\n")
    assert sync.stat().st_size > 6 << 20
    Y = synth.phenotypes(synth.genotype_matrix(64, n, "cpu", seed=73), n, k=1, seed=4)
    phen = tmp_path / "p.csv"
    with open(phen, "w") as fh:
        fh.write("#name,size,t1\n")
        for i in range(n):
            fh.write(f"P{i},20,{float(Y[i, 0])!r}\n")
    for analysis in ("ols_iter_with_kinship", "ols_iter"):
        base = [str(CLI), analysis, "-f", str(sync), "-p", str(phen), "--phen-delim", ",", "--phen-name-col", "0", "--phen-pool-size-col", "1",
                "--phen-value-col", "2", "--n-threads", "4", "--stream-chunk-mb", "2"]
        outs = {}
        for tag, extra in (("drop", {}), ("keep", {"PGH_KEEP_MAPPED": "1"})):
            o = tmp_path / f"{analysis}_{tag}.csv"
            r = subprocess.run(base + ["-o", str(o)], capture_output=True, text=True, env={**os.environ, **extra})
            assert r.returncode == 0, r.stderr
            outs[tag] = o.read_bytes()
        assert outs["drop"] == outs["keep"] and len(outs["drop"]) > 100000
        if analysis == "ols_iter":
            o = tmp_path / "one.csv"
            r = subprocess.run(base[:-2] + ["--stream-chunk-mb", "64", "-o", str(o)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert o.read_bytes() == outs["drop"]


def test_operator_interface_reference_unit_tests():
    """`apitest`: the reference's operator-level unit tests transcribed to the C++ mirror of its interface
    (host/operators.h: FilterStats, LocusCounts, LocusCountsAndPhenotypes, chisq / correlation / ols_iterate returning
    Option<String>): gwas/correlation_test.rs:136-182 and tables/chisq_test.rs:53-82 with their expected lines."""
    exe = ROOT / "poolgen_amd" / "csrc" / "apitest"
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "apitest: all checks passed" in r.stdout and "FAIL" not in r.stdout


def _sorted_fixture(tmp_path):
    lines = [l for l in (GOLD / "test.sync").read_text().splitlines() if not l.startswith("#")]
    lines.sort(key=lambda l: (l.split("\t")[0].encode(), int(l.split("\t")[1])))
    srt = tmp_path / "sorted.sync"
    srt.write_text("\n".join(lines) + "\n")
    return srt


@pytest.mark.parametrize("thr", ["0.5", "0.8"])
def test_multi_gpu_kinship_ranks(tmp_path, thr):
    """`--n-gpus`: the input is cut into one contiguous byte range per rank (own GPU context, own parser threads), the
    partial kinship sums are all-reduced, every rank sweeps its own pieces, the CSV is assembled in rank order.
    On the 1-GPU pool: (a) `--n-gpus 1` runs the REAL RCCL path (pg_comm_init_rank + pg_allreduce_sum_dev, one rank) and
    must reproduce the flag-less run byte for byte; (b) two and three ranks sharing GPU 0 (PGH_COMM=host: the rehearsal mode,
    sums added on the host in rank order because RCCL refuses two ranks per device) must give the same labels and, up to the
    summation order of the kinship (m = 0: nothing at all), the same numbers."""
    import os
    srt = _sorted_fixture(tmp_path)
    base = [str(CLI), "ols_iter_with_kinship", "-f", str(srt), "-p", str(GOLD / "test.csv"), "--phen-value-col", "2,3", "--n-threads", "4",
            "-x", thr]
    env = dict(os.environ, PGH_STREAM_CHUNK_BYTES="40000", PGH_TIMING="1")
    one, rccl1 = tmp_path / "one.csv", tmp_path / "rccl1.csv"
    r = subprocess.run(base + ["-o", str(one)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(base + ["-o", str(rccl1), "--n-gpus", "1"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "rank 0 (GPU 0" in r.stderr
    assert one.read_bytes() == rccl1.read_bytes() and one.stat().st_size > 100000
    la, va = _read_kinship_csv(one)
    for ranks, ids in ((2, "0,0"), (3, "0,0,0")):
        out = tmp_path / f"r{ranks}.csv"
        r = subprocess.run(base + ["-o", str(out), "--n-gpus", str(ranks), "--gpu-ids", ids], capture_output=True, text=True,
                           env=dict(env, PGH_COMM="host"))
        assert r.returncode == 0, r.stderr
        assert f"rank {ranks - 1} (GPU 0" in r.stderr
        lb, vb = _read_kinship_csv(out)
        assert la == lb
        assert np.allclose(va, vb, rtol=1e-10, atol=1e-11, equal_nan=True)
        if np.array_equal(va, vb, equal_nan=True):
            assert one.read_bytes() == out.read_bytes()
    # RCCL itself needs one GPU per rank; asking for two ranks on one device is refused with a clear message
    r = subprocess.run(base + ["-o", str(tmp_path / "x.csv"), "--n-gpus", "2", "--gpu-ids", "0,0"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "one GPU per rank" in r.stderr
    # ranks on GPUs the box does not have: refused before any rank thread exists (a rank that cannot open its device would
    # leave the others inside ncclCommInitRank for ever) -- an error within seconds, never a hang
    r = subprocess.run(base + ["-o", str(tmp_path / "z.csv"), "--n-gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr and not (tmp_path / "z.csv").exists()
    # an unsorted input cannot be split over ranks
    r = subprocess.run([a if a != str(srt) else str(GOLD / "test.sync") for a in base] + ["-o", str(tmp_path / "y.csv"), "--n-gpus", "2",
                       "--gpu-ids", "0,0"], capture_output=True, text=True, env=dict(env, PGH_COMM="host"))
    assert r.returncode != 0 and "sorted by chromosome and position" in r.stderr


@pytest.mark.parametrize("analysis,extra", [("ols_iter", ["--phen-value-col", "2,3"]), ("pearson_corr", ["--phen-value-col", "2"]),
                                            ("chisq_test", [])])
def test_multi_gpu_batch_operators(tmp_path, analysis, extra):
    """The per-locus operators over several ranks: no exchange, one part file per rank concatenated in rank order (the
    reference's one .tmp file per worker, sync.rs:794-870, :951-968) -- byte-identical to the single-rank file."""
    import os
    base = [str(CLI), analysis, "-f", str(GOLD / "test.sync"), "-p", str(GOLD / "test.csv"), "--n-threads", "4", *extra]
    env = dict(os.environ, PGH_STREAM_CHUNK_BYTES="50000")
    one = tmp_path / "one.csv"
    r = subprocess.run(base + ["-o", str(one)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    for ranks, ids in ((1, "0"), (2, "0,0"), (5, "0,0,0,0,0")):
        out = tmp_path / f"r{ranks}.csv"
        r = subprocess.run(base + ["-o", str(out), "--n-gpus", str(ranks), "--gpu-ids", ids], capture_output=True, text=True,
                           env=dict(env, PGH_COMM="host"))
        assert r.returncode == 0, r.stderr
        assert one.read_bytes() == out.read_bytes() and one.stat().st_size > 10000
        assert not list(tmp_path.glob("*.tmp"))
    r = subprocess.run([str(CLI), "fst", "-f", str(GOLD / "test.sync"), "-p", str(GOLD / "test.csv"), "--n-gpus", "2", "-o", str(tmp_path / "f.csv")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "runs on one GPU" in r.stderr


def test_cli_count_operators_on_error_bearing_text(oracle, tmp_path):
    """The three count operators through the CLI on sync TEXT that looks like real pool-seq data: 100 pools, every read misread with
    probability 0.5 % onto one of the other five columns, so nearly every locus carries reads of alleles the MAF filter drops and
    the reference recomputes the frequencies on the filtered counts (gwas/ols.rs:210-230).  Text fields identical to the oracle's
    CSV; ols_iter's 8-decimal mean frequency, beta and p within one unit of their grid / 1e-10 (its first pass is order-free from
    32 pools up); pearson_corr's full-precision mean as TEXT; chisq_test within its printing grid."""
    from poolgen_amd import synth
    n, L = 100, 1500
    counts = synth.sync_counts(L, n, "cpu", seed=61, error_rate=0.005).numpy().astype(np.uint64)
    counts[5::97, :10, :] = 0                                    # uncovered pools: dropped by the default --min-coverage-depth
    G = synth.genotype_matrix(64, n, "cpu", seed=61)
    Y = synth.phenotypes(G, n, k=2, seed=9)
    ps = [20.0] * n
    sync = tmp_path / "e.sync"
    with open(sync, "w") as fh:
        fh.write("#chr\tpos\tref\t" + "\t".join(f"p{i}" for i in range(n)) + "\n")
        for l in range(L):
            fh.write(f"chr{1 + l // 700}\t{1000 + 17 * l}\tN\t" + "\t".join(":".join(str(int(x)) for x in counts[l, i]) for i in range(n)) + "\n")
    phen = tmp_path / "e.csv"
    with open(phen, "w") as fh:
        fh.write("#name,size,t1,t2\n")
        for i in range(n):
            fh.write(f"P{i},20,{float(Y[i, 0])!r},{float(Y[i, 1])!r}\n")
    rows = [(f"chr{1 + l // 700}", 1000 + 17 * l, counts[l]) for l in range(L)]
    for maf in ("0.01", "0.001"):
        f = oracle.filt(True, 1, float(maf), 0.0)
        common = ["-f", sync, "-p", phen, "--phen-delim", ",", "--phen-name-col", "0", "--phen-pool-size-col", "1", "--min-allele-frequency", maf]
        out = tmp_path / f"ols_{maf}.csv"
        run_cli("ols_iter", *common, "--phen-value-col", "2,3", "--n-threads", 4, "-o", out)
        want = "#chr,pos,alleles,freq,phenotype,statistic,pvalue\n" + "".join(oracle.ols_iterate_csv(c, p, cnt, Y, ps, f) or "" for c, p, cnt in rows)
        assert want.count("\n") > L // 2
        compare_csv(out.read_text(), want, {3: 1.0000001e-8, 5: 1.0000001e-6, 6: 1e-10}, max_noise_rows=want.count("\n"), exempt=set())
        out = tmp_path / f"prs_{maf}.csv"
        run_cli("pearson_corr", *common, "--phen-value-col", "2,3", "--n-threads", 4, "-o", out)
        want = "#chr,pos,alleles,freq,phenotype,statistic,pvalue\n" + "".join(oracle.correlation_csv(c, p, cnt, Y, ps, f) or "" for c, p, cnt in rows)
        compare_csv(out.read_text(), want, {5: 1.0000001e-6, 6: 1e-10}, max_noise_rows=want.count("\n"), exempt=set())   # (column 3, the mean: text)
        out = tmp_path / f"chi_{maf}.csv"
        run_cli("chisq_test", *common, "--n-threads", 4, "-o", out)
        want = "#chr,pos,alleles,statistic,pvalue\n" + "".join(oracle.chisq_csv(c, p, cnt, ps, f) or "" for c, p, cnt in rows)
        compare_csv(out.read_text(), want, {3: 1.0000001e-6, 4: 1e-10}, max_noise_rows=want.count("\n"), exempt=set())
