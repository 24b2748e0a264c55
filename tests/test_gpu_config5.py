"""BASELINE config 5 (pileup2sync streamed -> ols_iter_with_kinship, pinned async H2D overlapped with compute) against the
ORACLE -- not against the product's own whole-file path: > 1 M pileup sites through the CLI's piece-wise path (>= 8 pieces:
parse of piece c + 1 || H2D + loader + partial kinship of piece c, G resident, one eigen rule, one sweep per piece), every
label compared exactly and every coefficient / p-value at 1e-10 with the oracle's pileup -> sync -> filter -> frequencies ->
kinship -> fits chain (base/pileup.rs:11-370, base/sync.rs:100-304, :972-1180, gwas/ols.rs:278-436).
The file is a block of 131 072 generated sites written 8 times under chromosome names a_chr1.. h_chr2 (sorted, as the streamed
path requires): the oracle converts and filters the block once (its per-line Python loop is the slow part) and then runs its
kinship and its fits over ALL 1 048 576 sites' columns.  Throughput and the GPU-idle share of the piece loop are printed and
left in gpurun_out/ (profiles/r02_config5_test.log is a copy)."""
import os
import random
import subprocess
import time
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CLI = ROOT / "poolgen_amd" / "csrc" / "poolgen"
N_POOLS = 8
BLOCK = 131072
COPIES = 8


def _block(rng):
    lines = []
    quals = "5:?DIJ"
    for l in range(BLOCK):
        ref = "ACGT"[l & 3]
        alt = "CGTA"[l & 3]
        f = min(0.95, max(0.05, rng.betavariate(0.6, 0.6)))
        parts = ["chr%d" % (1 + (l >= BLOCK // 2)), str(10 + 3 * l), ref]
        for _ in range(N_POOLS):
            cov = rng.randint(6, 18)
            fa = min(1.0, max(0.0, f + rng.gauss(0, 0.12)))
            b = rng.choices((alt, alt.lower(), ".", ","), weights=(fa, fa, 1.0 - fa, 1.0 - fa), k=cov)
            q = rng.choices(quals, k=cov)
            r = rng.random()
            if r < 0.10:                                 # one of the decorations the converter must step over or count
                i = rng.randrange(cov)
                if r < 0.03:
                    b[i] = "^F" + b[i]                   # read start + mapping quality
                elif r < 0.05:
                    b[i] = b[i] + "$"                    # read end
                elif r < 0.07:
                    b[i] = b[i] + "+2AG"                 # insertion after the base
                elif r < 0.08:
                    b[i] = b[i] + "-1N"                  # deletion after the base
                elif r < 0.09:
                    b[i] = "*"                           # deletion placeholder: counted as D
                else:
                    q[i] = "!"                           # below the quality limit: counted as N
            parts += [str(cov), "".join(b), "".join(q)]
        lines.append("\t".join(parts))
    return lines


def test_streamed_pileup_kinship_against_oracle(oracle, tmp_path):
    rng = random.Random(20261004)
    t0 = time.time()
    block = _block(rng)
    pile = tmp_path / "c5.pileup"
    with open(pile, "w", encoding="latin-1") as f:
        for c in range(COPIES):
            pre = "abcdefgh"[c] + "_"
            f.write("\n".join(pre + l for l in block) + "\n")
    phen = tmp_path / "phen.csv"
    yr = np.random.default_rng(5).normal(size=N_POOLS)
    phen.write_text("#pool,size,trait\n" + "".join("pool%d,%d,%r\n" % (i, 20 + i, float(yr[i])) for i in range(N_POOLS)))
    fsize = pile.stat().st_size
    t_gen = time.time() - t0
    out = tmp_path / "c5.csv"
    env = dict(os.environ, PGH_STREAM_CHUNK_BYTES=str(fsize // 12 + 1), PGH_TIMING="1")
    t0 = time.time()
    r = subprocess.run([str(CLI), "ols_iter_with_kinship", "-f", str(pile), "-p", str(phen), "--phen-value-col", "2", "--n-threads", "16",
                        "-o", str(out)], capture_output=True, text=True, env=env)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr
    timing = [l for l in r.stderr.splitlines() if l.startswith("poolgen:")]
    rank = [l for l in timing if "rank 0" in l][0]
    import re
    pieces = tuple(int(x) for x in re.search(r"pieces (\d+)\.\.(\d+)", rank).groups())
    assert pieces[1] - pieces[0] >= 8
    t_wait, t_host, t_dev = (float(x) for x in re.search(r"parser ([\d.]+) s, host bookkeeping ([\d.]+) s, copies \+ device ([\d.]+) s", rank).groups())
    idle = 1.0 - t_dev / max(t_wait + t_host + t_dev, 1e-9)

    # ---- the oracle's chain on the block --------------------------------------------------------------------------------
    t0 = time.time()
    ps = np.array([20.0 + i for i in range(N_POOLS)]); ps = ps / ps.sum()
    f = oracle.filt()
    labels, cols = [], []
    for line in block:
        rc, sync = oracle.pileup_to_sync(line, ps)
        assert rc >= 0
        if rc == 0:
            continue
        n, chrom, pos, counts = oracle.parse_sync_line(sync.rstrip("\n"))
        assert n == N_POOLS
        res = oracle.filter_locus(counts, ps, f)
        if res is None:
            continue
        ids, fc = res
        fr = oracle.to_frequencies(fc)
        for j, a in enumerate(ids):
            labels.append((chrom, pos, "ATCGND"[a])); cols.append(fr[:, j])
    Gb = np.array(cols)
    pb = len(cols)
    assert pb > BLOCK // 2       # (about half of the generated sites pass the converter's and the loader's filters, two columns each)
    G = np.tile(Gb, (COPIES, 1))
    ref = oracle.ols_with_covariate(G, yr.reshape(-1, 1), 0.75)
    t_or = time.time() - t0
    assert ref["m"] == 0

    # ---- compare: header, the shifted labels (gwas/ols.rs:421-425), every number ---------------------------------------------
    with open(out) as fh:
        assert fh.readline() == "#chr,pos,alleles,phenotype,statistic,pvalue\n"
        got = fh.read().splitlines()
    p = pb * COPIES
    assert len(got) == p
    beta = np.empty(p); pv = np.empty(p)
    want_lab = [("intercept", 0, "intercept")]
    for c in range(COPIES):
        pre = "abcdefgh"[c] + "_"
        want_lab += [(pre + ch, po, al) for ch, po, al in labels]
    for i, line in enumerate(got):
        fa = line.split(",")
        assert (fa[0], int(fa[1]), fa[2], fa[3]) == (want_lab[i][0], want_lab[i][1], want_lab[i][2], "Pheno_0"), (i, line)
        beta[i] = float(fa[4]); pv[i] = float(fa[5])
    rb, rp = ref["beta"][:, 0], ref["pval"][:, 0]
    nan_ref = np.isnan(rb)
    # the product flags s_gg <= 1e-12 g'g as NaN where the reference prints noise unless its LU hits an exact zero
    extra_nan = np.isnan(beta) & ~nan_ref
    assert extra_nan.mean() < 0.01 and not np.any(nan_ref & ~np.isnan(beta))
    ok = ~np.isnan(beta)
    assert np.allclose(beta[ok], rb[ok], rtol=1e-10, atol=1e-10 * float(np.max(np.abs(rb[ok]))))
    assert np.max(np.abs(pv[ok] - rp[ok])) <= 1e-10
    msg = (f"config-5 test: {fsize / 1e6:.0f} MB of mpileup text, {BLOCK * COPIES} sites x {N_POOLS} pools -> {p} allele columns, "
           f"{pieces[1] - pieces[0]} pieces; CLI wall {wall:.2f} s = {fsize / wall / 1e9:.2f} GB/s of text; piece loop: waited for the parser "
           f"{t_wait:.2f} s, host {t_host:.2f} s, copies + device {t_dev:.2f} s => GPU idle {100 * idle:.0f} % of the loop; "
           f"{int(extra_nan.sum())} columns flagged NaN by the product only; (generation {t_gen:.0f} s, oracle {t_or:.0f} s)")
    print("\n" + msg + "\n" + "\n".join(timing))
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "r02_config5_test.log").write_text(msg + "\n" + "\n".join(timing) + "\n")


def test_streamed_sync_kinship_200_pools(oracle, tmp_path):
    """The streamed path at BASELINE's pool count: 200 pools x 2 097 152 sites of sync text (3.6 GB) in >= 8 pieces through
    `poolgen ols_iter_with_kinship`, the file being 32 copies of one block of 65 536 sites under chromosome names that keep it
    sorted.  The oracle's chain (parse -> filter -> frequencies -> kinship -> eigen rule -> fits) runs on the block: the kinship of
    32 copies IS the block's (K = sum g g' / p), so every copy's coefficients and p-values must equal the block's at 1e-10, and
    every label of the 4 M-row file is checked.  About 3 % of the sites carry a third allele with a few reads (dropped by the
    MAF filter WITH reads: the second pass over the survivors).  Text GB/s and the GPU-idle share of the piece loop go to
    gpurun_out/r03_config5_n200.log (profiles/ holds a copy)."""
    import re
    n, block, copies = 200, 65536, 32
    rng = np.random.default_rng(20261005)
    t0 = time.time()
    base = np.clip(rng.beta(0.5, 0.5, size=(block, 1)), 0.03, 0.97)
    d = rng.poisson(40, size=(block, n)) + 8
    a = rng.binomial(d, np.clip(base + rng.normal(0, 0.08, size=(block, n)), 0, 1))
    err = (rng.random(size=(block, 1)) < 0.03) * (rng.random(size=(block, n)) < 0.02)      # a sequencing-error allele in a few pools
    body = []
    for l in range(block):
        body.append("%d\tN\t" % (100 + 7 * l) + "\t".join("%d:%d:%d:0:0:0" % (a[l, i], d[l, i] - a[l, i], 1 if err[l, i] else 0) for i in range(n)))
    names = ["c%02d" % c for c in range(copies)]
    sync = tmp_path / "c5_200.sync"
    with open(sync, "w") as f:
        for nm in names:
            f.write("\n".join(nm + "\t" + b for b in body) + "\n")
    fsize = sync.stat().st_size
    yr = rng.normal(size=n)
    phen = tmp_path / "phen.csv"
    phen.write_text("#pool,size,trait\n" + "".join("pool%d,%d,%r\n" % (i, 20 + (i % 7), float(yr[i])) for i in range(n)))
    t_gen = time.time() - t0
    out = tmp_path / "c5_200.csv"
    env = dict(os.environ, PGH_STREAM_CHUNK_BYTES=str(fsize // 12 + 1), PGH_TIMING="1")
    t0 = time.time()
    r = subprocess.run([str(CLI), "ols_iter_with_kinship", "-f", str(sync), "-p", str(phen), "--phen-value-col", "2", "--n-threads", "16",
                        "-o", str(out)], capture_output=True, text=True, env=env)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr
    timing = [l for l in r.stderr.splitlines() if l.startswith("poolgen:")]
    rank = [l for l in timing if "rank 0" in l][0]
    pieces = tuple(int(x) for x in re.search(r"pieces (\d+)\.\.(\d+)", rank).groups())
    assert pieces[1] - pieces[0] >= 8
    t_wait, t_host, t_dev = (float(x) for x in re.search(r"parser ([\d.]+) s, host bookkeeping ([\d.]+) s, copies \+ device ([\d.]+) s", rank).groups())
    idle = 1.0 - t_dev / max(t_wait + t_host + t_dev, 1e-9)

    # ---- the oracle's chain on the block ------------------------------------------------------------------------------------
    t0 = time.time()
    ps = np.array([20.0 + (i % 7) for i in range(n)]); ps = ps / ps.sum()
    flt = oracle.filt()
    labels, cols, second = [], [], 0
    for l in range(block):
        nn, chrom, pos, counts = oracle.parse_sync_line(names[0] + "\t" + body[l])
        assert nn == n
        res = oracle.filter_locus(counts, ps, flt)
        if res is None:
            continue
        ids, fc = res
        second += int(err[l].any() and 2 not in ids)
        fr = oracle.to_frequencies(fc)
        for j, al in enumerate(ids):
            labels.append((pos, "ATCGND"[al])); cols.append(fr[:, j])
    Gb = np.array(cols)
    pb = len(cols)
    assert pb > block and second > 500      # two columns per surviving site; the error alleles were dropped with reads
    ref = oracle.ols_with_covariate(Gb, yr.reshape(-1, 1), 0.75)
    t_or = time.time() - t0

    # ---- every row of the file: labels exactly (shifted by one, gwas/ols.rs:421-425), numbers against the block's -------------
    rb, rp = ref["beta"][:, 0], ref["pval"][:, 0]
    p = pb * copies
    beta = np.empty(p); pv = np.empty(p)
    with open(out) as fh:
        assert fh.readline() == "#chr,pos,alleles,phenotype,statistic,pvalue\n"
        i = 0
        for line in fh:
            fa = line.rstrip("\n").split(",")
            if i == 0:
                assert fa[:4] == ["intercept", "0", "intercept", "Pheno_0"]
            else:
                c, q = divmod(i - 1, pb)
                assert fa[0] == names[c] and int(fa[1]) == labels[q][0] and fa[2] == labels[q][1] and fa[3] == "Pheno_0", (i, line)
            beta[i] = float(fa[4]); pv[i] = float(fa[5])
            i += 1
    assert i == p
    want_b, want_p = np.tile(rb, copies), np.tile(rp, copies)
    ok = ~np.isnan(beta)
    extra_nan = np.isnan(beta) & ~np.isnan(want_b)
    assert extra_nan.mean() < 0.01 and not np.any(np.isnan(want_b) & ok)
    assert np.allclose(beta[ok], want_b[ok], rtol=1e-10, atol=1e-10 * float(np.max(np.abs(rb[~np.isnan(rb)]))))
    assert np.max(np.abs(pv[ok] - want_p[ok])) <= 1e-10
    msg = (f"config-5 test at {n} pools: {fsize / 1e9:.2f} GB of sync text, {block * copies} sites -> {p} allele columns, {pieces[1] - pieces[0]} pieces, "
           f"{second * copies} sites through the second pass; CLI wall {wall:.2f} s = {fsize / wall / 1e9:.2f} GB/s of text; piece loop: waited for the "
           f"parser {t_wait:.2f} s, host {t_host:.2f} s, copies + device {t_dev:.2f} s => GPU idle {100 * idle:.0f} % of the loop; "
           f"{int(extra_nan.sum())} columns flagged NaN by the product only; (generation {t_gen:.0f} s, oracle on the block {t_or:.0f} s)")
    print("\n" + msg + "\n" + "\n".join(timing))
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "r03_config5_n200.log").write_text(msg + "\n" + "\n".join(timing) + "\n")
