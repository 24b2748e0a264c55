"""CPU tests of the CLI's host logic (C++): Rust-compatible formatting, phenotype parser, sync
parser and the locus filter/loader, against the oracle and the reference's literals."""
import json
import random
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
HC = ROOT / "poolgen_amd" / "csrc" / "hostcheck"
GOLD = Path(__file__).parent / "golden"
LIT = json.loads((GOLD / "reference_literals.json").read_text())


def run(*args, stdin=None):
    if not HC.exists():
        subprocess.check_call(["make", "-C", str(HC.parent), "hostcheck", "-s"])
    return subprocess.run([str(HC), *map(str, args)], input=stdin, capture_output=True, text=True, check=True).stdout


def test_formatting_matches_oracle(oracle):
    random.seed(3)
    xs = [0.3, 4.0, 1e-7, 1.5e22, 0.1 + 0.2, -0.0, 0.0, 5e-324, 0.420000012435, 0.690000012435, float("nan"), float("inf")]
    xs += [random.choice([random.random(), random.gauss(0, 50), random.random() * 1e-9, random.random() * 1e12,
                          round(random.random(), 3), float(random.randint(-9, 9)) + 0.5]) for _ in range(20000)]
    nds = [random.choice([4, 6, 7, 8, 12]) for _ in xs]
    out = run("fmt", stdin="".join(f"{x!r} {nd}\n" for x, nd in zip(xs, nds))).splitlines()
    assert len(out) == len(xs)
    for x, nd, line in zip(xs, nds, out):
        d, r = line.split(" ")
        assert d == oracle.fmt(x), x
        assert r == oracle.round_own(x, nd), (x, nd)


def test_phen_parser_reference_literals():
    g = LIT["phen"]  # base/phen.rs:221-236
    rows = [l.split(" ") for l in run("phen", GOLD / "test.csv", ",", 0, 1, "2,3").splitlines()]
    assert [r[0] for r in rows] == g["pool_names"]
    assert [float(r[1]) for r in rows] == g["pool_sizes"]
    assert [float(r[2]) for r in rows] == g["phen_matrix_by_trait"][0]
    assert [float(r[3]) for r in rows] == g["phen_matrix_by_trait"][1]


def test_phen_parser_missing_values(tmp_path):
    f = tmp_path / "p.tsv"
    f.write_text("#h\nA\t10\tNA\t1.5\r\nB\t30\t 2.5 \t\nC\t60\tnan\t-1e3\n")
    rows = [l.split(" ") for l in run("phen", f, "\t", 0, 1, "2,3").splitlines()]
    assert [r[1] for r in rows] == ["0.1", "0.3", "0.6"]          # phen.rs:83-84
    assert [r[2] for r in rows] == ["NaN", "2.5", "NaN"] and [r[3] for r in rows] == ["1.5", "NaN", "-1000"]


def test_sync_parser_matches_oracle_and_keeps_file_order(oracle):
    """parse_sync_file (threads over byte ranges, helpers.rs:74-91) = lparse per line (sync.rs:100-156), in file
    order.  Filter / frequencies / column layout of the loader run on the GPU (tests/test_gpu_locus_ops.py)."""
    lines = [l for l in (GOLD / "test.sync").read_text().splitlines() if not l.startswith("#")]
    for threads in (1, 2, 3, 7):
        out = run("parse", GOLD / "test.sync", threads).splitlines()
        L, n = map(int, out[0].split())
        assert (L, n) == (len(lines), 5) == (6674, 5)
        for line, got in zip(lines, out[1:]):
            _, chrom, pos, counts = oracle.parse_sync_line(line)
            parts = got.split(" ")
            assert parts[0] == chrom and int(parts[1]) == pos
            assert [int(x) for x in parts[2:]] == np.asarray(counts).reshape(-1).tolist()


def _parse_text(tmp_path, text, threads=2):
    f = tmp_path / "t.sync"
    f.write_bytes(text.encode())
    r = subprocess.run([str(HC), "parse", str(f), str(threads)], capture_output=True, text=True)
    return r.returncode, r.stdout.splitlines(), r.stderr


def test_sync_parser_edge_cases(tmp_path):
    """Comment lines and lines whose position is not an integer are skipped (both ErrorKind::Other -> `continue`,
    sync.rs:111-128, :829-846); CRLF is stripped (:104-109); only the first six counts of a pool are used (:141-146);
    a last line without newline still counts; malformed counts and ragged lines are errors."""
    run("fmt", stdin="")  # builds hostcheck if needed
    rc, out, _ = _parse_text(tmp_path, "#c\nchr1\t10\tN\t1:2:3:4:5:6\t7:8:9:10:11:12\r\nchr1\tx\tN\t1:1:1:1:1:1\t1:1:1:1:1:1\n"
                                      "chr2\t+30\tA\t0:0:0:0:0:4294967295:77\t1:0:0:0:0:0")
    assert rc == 0
    assert out == ["2 2", "chr1 10 1 2 3 4 5 6 7 8 9 10 11 12", "chr2 30 0 0 0 0 0 4294967295 1 0 0 0 0 0"]
    for bad in ("chr1\t1\tN\t1:2:3:4:5\t1:2:3:4:5:6\n",            # five counts
                "chr1\t1\tN\t1:2:3:4:5:x\t1:2:3:4:5:6\n",          # not an integer
                "chr1\t1\tN\t1:2:3:4:5:6\t1:2:3:4:5:6\nchr1\t2\tN\t1:2:3:4:5:6\n",   # ragged
                "chr1\t1\tN\t1:2:3:4:5:4294967296\n",              # beyond the u32 of the device layout
                "chr1\t1\tN\t1:2:3:4:5:6\n\nchr1\t2\tN\t1:2:3:4:5:6\n"):            # empty line (reference panics)
        rc, _, err = _parse_text(tmp_path, bad)
        assert rc != 0 and "hostcheck:" in err, bad
    rc, out, _ = _parse_text(tmp_path, "# only comments\n#\n")
    assert rc == 0 and out == ["0 0"]


def test_seeded_k_split_matches_restatement():
    """The CV harness's folds (gp/cv.rs:15-49 with a seeded shuffle): C++ against the restatement the GPU test uses."""
    from test_gpu_cv import SplitMix64, k_split
    for n, k, seed in ((36, 3, 11), (45, 10, 3), (200, 10, 42), (19, 5, 1), (25, 10, 7)):
        out = run("ksplit", n, k, seed).split()
        g, kk = k_split(n, k, SplitMix64(seed).permutation(n))
        assert [int(x) for x in out] == [kk] + g
    r = subprocess.run([str(HC), "ksplit", "10", "10", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "number of splits" in r.stderr


def test_sync_parser_16_bit_counts(tmp_path):
    """The compact form of the counts (what the kinship paths copy to the GPU): same numbers, and a count above 65535
    anywhere in the input sends the whole batch back to 32 bits."""
    small = "".join("chr%d\t%d\tN\t%d:%d:0:0:0:%d\t7:65535:0:1:0:0\t0:0:0:0:0:0\n" % (1 + i // 40, 10 + i, i, 3 * i, i % 5) for i in range(100))
    p = tmp_path / "s.sync"; p.write_text(small)
    for threads in (1, 3):
        a = run("parse", p, threads).splitlines()
        b = run("parse", p, threads, 16).splitlines()
        assert b[0] == a[0] + " 16" and b[1:] == a[1:]
    big = small + "chr9\t5\tN\t1:2:3:4:5:6\t65536:0:0:0:0:0\t1:1:1:1:1:1\n" + small.replace("chr", "chs")
    p.write_text(big)
    for threads in (1, 4):
        a = run("parse", p, threads).splitlines()
        b = run("parse", p, threads, 16).splitlines()
        assert b[0] == a[0] + " 32" and b[1:] == a[1:]


def test_numbers_are_read_like_rust_parse():
    """Flag values and phenotype numbers: what Rust's `str::parse::<f64 / usize>()` accepts (main.rs flag types, phen.rs:60-78)
    and nothing else -- strtod / stoi took hex floats, trailing garbage ("2x" -> 2) and wrapped negatives."""
    run("fmt", stdin="")
    ok = {("f64", "1.5"): "1.5", ("f64", "-2"): "-2", ("f64", "1e3"): "1000", ("f64", ".5"): "0.5", ("f64", "5."): "5",
          ("f64", "+0.25"): "0.25", ("f64", "inf"): "inf", ("f64", "-Infinity"): "-inf", ("f64", "NaN"): "NaN", ("f64", "1E-2"): "0.01",
          ("u64", "0"): "0", ("u64", "+17"): "17", ("u64", "18446744073709551615"): "18446744073709551615",
          ("i64", "-5"): "-5", ("i64", "42"): "42"}
    for (kind, text), want in ok.items():
        assert run("num", kind, text).strip() == "ok " + want, (kind, text)
    bad = [("f64", ""), ("f64", "0x10"), ("f64", "0x1p3"), ("f64", "1.5x"), ("f64", " 1"), ("f64", "1 "), ("f64", "e5"), ("f64", "."),
           ("f64", "1e"), ("f64", "--1"), ("f64", "infin"), ("u64", "-1"), ("u64", "2x"), ("u64", "1e3"), ("u64", ""), ("u64", "1.0"),
           ("u64", "18446744073709551616"), ("i64", "9223372036854775808"), ("i64", "-+1"), ("i64", "3 ")]
    for kind, text in bad:
        assert run("num", kind, text).strip() == "reject", (kind, text)
