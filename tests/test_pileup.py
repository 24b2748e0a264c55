"""pileup2sync (base/pileup.rs): the CLI's converter against the reference's own literal and, on generated
lines, against the oracle's literal restatement -- kept / dropped / fatal decisions and the sync text are identical."""
import json
import random
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HC = ROOT / "poolgen_amd" / "csrc" / "hostcheck"
LIT = json.loads((Path(__file__).parent / "golden" / "reference_literals.json").read_text())


def hostcheck(*args):
    subprocess.check_call(["make", "-C", str(HC.parent), "hostcheck", "-s"])
    return subprocess.run([str(HC), *map(str, args)], capture_output=True, text=True, check=True).stdout


def test_reference_literal(tmp_path):
    g = LIT["pileup"]; f = g["filter"]
    p = tmp_path / "a.pileup"; p.write_text(g["line"] + "\n")
    ps = ",".join(map(str, f["pool_sizes"]))
    out = hostcheck("pileuplines", p, 1, 1.0, 1, 1.0, 0.0, ps).splitlines()
    assert out == ["K Chromosome1\t456527\tC\t" + "\t".join(":".join(map(str, r)) for r in g["counts_ATCGDN"])]
    out = hostcheck("pileuplines", p, int(f["remove_ns"]), f["max_base_error_rate"], f["min_coverage_depth"],
                    f["min_coverage_breadth"], f["min_allele_frequency"], ps).splitlines()
    cov = [sum(map(int, x.split(":"))) for x in out[0][2:].split("\t")[3:]]
    assert cov == g["filtered_coverages"]


def _random_line(rng, n, weird):
    ref = rng.choice("ACGTNacgt")
    fields = ["chr%d" % rng.randint(1, 3), str(rng.randint(1, 10**7)), ref]
    for _ in range(n):
        cov = rng.choice([0, 0, 1, 2, 3, 5, 8, 13]) if weird else rng.randint(1, 9)
        bases, quals = [], []
        for _ in range(cov):
            r = rng.random()
            if r < 0.08:
                bases.append("^" + rng.choice("!~+-$^5J"))          # read start + mapping quality (any character)
            b = rng.choice(".,.,.,ACGTacgtNn*") if rng.random() > 0.02 else rng.choice("XxRy><")
            bases.append(b)
            if rng.random() < 0.06:
                bases.append("$")
            if rng.random() < 0.10:                                   # indel after the base
                k = rng.choice([1, 2, 3, 10, 12]) if rng.random() > 0.05 else 0
                seq = "".join(rng.choice("ACGTNacgtn*+-^$.,") for _ in range(k if k else 1))
                bases.append(rng.choice("+-") + (str(k) if k else "0" + str(len(seq))) + seq)
            quals.append(chr(rng.choice([33, 34, 40, 45, 50, 53, 55, 60, 70, 74]) if rng.random() > 0.002 else rng.choice([31, 32])))
        b, q = "".join(bases), "".join(quals)
        if cov == 0:
            b, q = "*", "*"
        if weird and rng.random() < 0.005:
            q = q[:-1] if q else "J"                                  # coverage / qualities mismatch -> fatal
        if weird and rng.random() < 0.004:
            b = b + "+x"                                              # indel length is not a digit -> fatal
        fields += [str(cov), b, q]
    if weird and rng.random() < 0.02:
        fields = fields[:-1]                                          # ragged pool
    if weird and rng.random() < 0.02:
        fields[1] = "12a"
    return "\t".join(fields)


def test_generated_lines_match_oracle(tmp_path, oracle):
    rng = random.Random(20251003)
    for n, remove_ns, max_err, depth, breadth, maf, weird, keep_lc in [
            (5, True, 0.01, 1, 1.0, 0.001, False, False), (3, False, 0.005, 2, 0.5, 0.05, True, False),
            (8, True, 0.0005, 1, 0.75, 0.0, True, False), (2, True, 1.0, 3, 1.0, 0.05, True, False),
            (4, True, 0.01, 1, 1.0, 0.001, True, True), (3, False, 0.01, 1, 0.6, 0.02, True, True)]:   # --keep-lowercase-reference (:280-299)
        ps = [1.0 / n] * n
        lines = [_random_line(rng, n if rng.random() > 0.01 else n + 1, weird) for _ in range(1500)]
        p = tmp_path / "g.pileup"; p.write_text("\n".join(lines) + "\n", encoding="latin-1")
        got = hostcheck("pileuplines", p, int(remove_ns) + 2 * int(keep_lc), max_err, depth, breadth, maf, ",".join(map(repr, ps))).splitlines()
        assert len(got) == len(lines)
        kinds = {"K": 0, "D": 0, "E": 0}
        for line, g in zip(lines, got):
            rc, text = oracle.pileup_to_sync(line, ps, remove_ns, max_err, depth, breadth, maf, keep_lowercase_reference=keep_lc)
            want = ("K " + text.rstrip("\n")) if rc > 0 else ("D" if rc == 0 else "E")
            assert g == want, (line, g, want, rc)
            kinds[g[0]] += 1
        assert kinds["K"] > 20 and (not weird or (kinds["D"] > 0 and kinds["E"] > 0)), kinds


def test_file_conversion_threads_header_and_overwrite(tmp_path, oracle):
    rng = random.Random(7)
    n = 4
    ps = [0.25] * n
    lines = [_random_line(rng, n, False) for _ in range(997)]
    p = tmp_path / "f.pileup"; p.write_text("\n".join(lines))          # no trailing newline
    want = "#chr\tpos\tref\ta\tb\tc\td\n" + "".join(oracle.pileup_to_sync(l, ps)[1] for l in lines)
    for threads in (1, 3, 8):
        out = tmp_path / f"o{threads}.sync"
        msg = hostcheck("pileup2sync", p, out, threads, "a,b,c,d", 1, 0.01, 1, 1.0, 0.001, ",".join(map(repr, ps)))
        assert out.read_text() == want and msg.split()[0] == str(want.count("\n") - 1)
    r = subprocess.run([str(HC), "pileup2sync", str(p), str(out), "2", "a,b,c,d", "1", "0.01", "1", "1.0", "0.001", "0.25,0.25,0.25,0.25"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Unable to create file" in r.stderr   # create_new: refuses to overwrite (pileup.rs:511-517)


def test_cli_pileup2sync_subcommand(tmp_path):
    """`poolgen pileup2sync -f x.pileup -p phen.csv` (main.rs:212-225): header from the phenotype file's pool names,
    default output name <input without extension>-<time>.sync, no GPU needed."""
    exe = ROOT / "poolgen_amd" / "csrc" / "poolgen"
    if not exe.exists():
        subprocess.check_call(["make", "-C", str(exe.parent), "poolgen", "-s"])
    g = LIT["pileup"]
    p = tmp_path / "x.y.pileup"; p.write_text(g["line"] + "\n")
    phen = Path(__file__).parent / "golden" / "test.csv"
    r = subprocess.run([str(exe), "pileup2sync", "-f", str(p), "-p", str(phen), "--max-base-error-rate", "0.005",
                        "--min-allele-frequency", "0.0", "--n-threads", "2"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    out = Path(r.stdout.strip())
    assert out.name.startswith("x.y-") and out.suffix == ".sync"
    lines = out.read_text().splitlines()
    assert lines[0] == "#chr\tpos\tref\tG1\tG2\tG3\tG4\tG5"
    assert lines[1].split("\t")[-1] == "0:1:5:0:0:0" and len(lines) == 2


def test_cli_keep_lowercase_reference_flag(tmp_path, oracle):
    """--keep-lowercase-reference reaches the converter (pileup.rs:280-299): on lines whose reference allele is lower
    case the sync counts differ from the default, and both equal the oracle's."""
    exe = ROOT / "poolgen_amd" / "csrc" / "poolgen"
    rng = random.Random(99)
    lines = [l for l in (_random_line(rng, 5, False) for _ in range(600)) if l.split("\t")[2] in "acgt"][:120]
    assert len(lines) > 50
    p = tmp_path / "lc.pileup"; p.write_text("\n".join(lines) + "\n", encoding="latin-1")
    phen = Path(__file__).parent / "golden" / "test.csv"
    ps = [0.2] * 5
    outs = {}
    for flag in (False, True):
        out = tmp_path / f"o{int(flag)}.sync"
        args = [str(exe), "pileup2sync", "-f", str(p), "-p", str(phen), "--min-allele-frequency", "0.0", "-o", str(out)]
        r = subprocess.run(args + (["--keep-lowercase-reference"] if flag else []), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[flag] = out.read_text().splitlines()[1:]
        want = [oracle.pileup_to_sync(l, ps, True, 0.01, 1, 1.0, 0.0, keep_lowercase_reference=flag)[1].rstrip("\n") for l in lines]
        assert outs[flag] == [w for w in want if w]
    assert outs[False] != outs[True]
