"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the header
declares, and refuses to compute without a GPU (no silent fallback)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def header_functions():
    text = (ROOT / "include" / "poolgen_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(native):
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(native, name), f"{name} declared in include/poolgen_hip.h but not exported"


def test_python_binding_covers_the_header():
    from poolgen_amd._native import SIGNATURES
    assert sorted(SIGNATURES) == header_functions()


def test_integration_doc_binds_every_entry_point():
    """INTEGRATION.md's `extern "C"` block is the reference-side binding: it must declare every function of the header."""
    text = (ROOT / "INTEGRATION.md").read_text()
    missing = [n for n in header_functions() if f"fn {n}(" not in text]
    assert not missing, missing


def test_no_product_code_touches_the_oracle():
    """The product (poolgen_amd/, include/) must never include, link, import or call oracle/."""
    bad = re.compile(r"poolgen_oracle\.h|liboracle|\borc_[a-z_]+\s*\(|oracle_lib|import\s+oracle|from\s+oracle|oracle/")
    for path in list((ROOT / "poolgen_amd").rglob("*")) + [ROOT / "include" / "poolgen_hip.h"]:
        if path.is_file() and path.suffix in {".py", ".hip", ".cpp", ".h"} or path.name == "Makefile":
            m = bad.search(path.read_text())
            assert m is None, f"{path}: {m.group(0)}"
    # and the built library has no dependency on it either
    out = subprocess.run(["ldd", str(ROOT / "poolgen_amd" / "csrc" / "libpoolgen_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_fails_loudly_without_gpu(native):
    ctx = C.c_void_p()
    rc = native.pg_create(C.byref(ctx), 0, None)
    assert rc == -3 and not ctx.value
    assert b"no CPU fallback" in native.pg_last_error(None)
    from poolgen_amd import Engine, NativeError
    with pytest.raises(NativeError):
        Engine(0)


def test_host_eig_matches_lapack(native):
    rng = np.random.default_rng(11)
    for n in (2, 5, 37, 200):
        G = rng.random((600, n)) * 0.6 + rng.random((600, 1)) * 0.4
        K = G.T @ G / 600
        ev = np.empty(n); V = np.empty((n, n))
        assert native.pg_host_sym_eig(K.ctypes.data, n, ev.ctypes.data, V.ctypes.data) == 0
        w = np.linalg.eigvalsh(K)[::-1]
        assert np.allclose(ev, w, rtol=0, atol=1e-13 * w[0])
        assert np.abs(K @ V - V * ev).max() < 1e-12 * w[0]
        assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12
        ev2 = np.empty(n)
        assert native.pg_host_sym_eig(K.ctypes.data, n, ev2.ctypes.data, None) == 0
        assert np.allclose(ev2, ev, rtol=0, atol=1e-13 * w[0])


def test_host_rule_and_pvalue_match_oracle(native, oracle):
    rng = np.random.default_rng(5)
    for _ in range(50):
        ev = np.sort(rng.random(12))[::-1].copy()
        thr = float(rng.random())
        assert native.pg_host_n_eigenvecs(ev.ctypes.data, 12, thr) == oracle.n_eigenvecs(ev, thr)
    # finite series (device algorithm) vs the statrs continued fraction (oracle).  statrs' own
    # Lanczos ln_gamma limits it to ~1e-13 relative; the contract is 1e-10 absolute.
    for df in (1, 2, 3, 4, 7, 98, 99, 199, 498, 499):
        for t in (1e-9, 1e-3, 0.1, 0.5, 1.0, 2.0, 3.5, 6.0, 12.0, 40.0, 1e3, 1e8):
            want = 2.0 * (1.0 - oracle.lib.orc_students_t_cdf(t, float(df)))
            got = native.pg_host_t_two_sided_p(t, df)
            assert abs(got - want) < 2e-12, (df, t, got, want)


def test_host_pinv_matches_numpy(native):
    rng = np.random.default_rng(2)
    X = rng.random((6, 40))
    A = X @ X.T
    A[:, 5] = A[:, 4]; A[5, :] = A[4, :]  # rank deficient
    out = np.empty_like(A)
    assert native.pg_host_pinv_sym(A.ctypes.data, 6, out.ctypes.data) == 0
    assert np.allclose(out, np.linalg.pinv(A), atol=1e-9)


def test_host_top_eigenvectors_match_lapack(native):
    """pg_host_sym_eig_top: all eigenvalues, the m leading vectors by inverse iteration (incl. clustered and repeated
    eigenvalues, where it must either orthogonalise or fall back to the full solve)."""
    import ctypes as C
    rng = np.random.default_rng(11)
    cases = []
    G = rng.random((120, 3000)); cases.append((G @ G.T / 3000, 6))                       # kinship-like: one dominant value
    Q, _ = np.linalg.qr(rng.normal(size=(60, 60)))
    lam = np.array([5.0, 5.0, 5.0 - 1e-9, 3.0, 3.0 + 1e-13] + list(np.linspace(2, 0.1, 55)))
    cases.append(((Q * lam) @ Q.T, 5))                                                   # repeated / nearly repeated values
    cases.append((np.diag(np.arange(1.0, 41.0)), 4))                                     # already diagonal
    for A, m in cases:
        A = np.ascontiguousarray((A + A.T) / 2)
        n = A.shape[0]
        ev = np.empty(n); V = np.empty((n, m))
        assert native.pg_host_sym_eig_top(A.ctypes.data, n, m, ev.ctypes.data, V.ctypes.data) == 0
        w = np.linalg.eigvalsh(A)[::-1]
        assert np.allclose(ev, w, rtol=0, atol=1e-12 * abs(w[0]))
        assert np.allclose(V.T @ V, np.eye(m), atol=1e-9)
        assert np.max(np.abs(A @ V - V * ev[:m])) <= 1e-10 * abs(w[0])


def test_host_sliding_windows_match_reference_literals_and_oracle(native, oracle):
    """define_sliding_windows (base/helpers.rs:294-403): the reference's two literal cases (:541-583), then random
    chromosomes / positions / window parameters against the oracle's restatement."""
    import json
    lit = json.loads((ROOT / "tests" / "golden" / "reference_literals.json").read_text())["popgen"]["windows"]

    def run(chrom, pos, w, s, m):
        ids = {}
        ch = np.array([ids.setdefault(c, len(ids)) for c in chrom], dtype=np.int32)
        po = np.ascontiguousarray(pos, dtype=np.uint64)
        head = np.empty(len(ch), dtype=np.int64); tail = np.empty(len(ch), dtype=np.int64)
        nw = native.pg_host_sliding_windows(ch.ctypes.data, po.ctypes.data, len(ch), w, s, m, head.ctypes.data, tail.ctypes.data)
        return head[:nw].tolist(), tail[:nw].tolist()

    for c in lit:
        assert run(c["chr"], c["pos"], c["window_size_bp"], c["window_slide_size_bp"], c["min_loci_per_window"]) == (c["head"], c["tail"])
    rng = np.random.default_rng(4)
    for _ in range(200):
        l = int(rng.integers(1, 400))
        chrom = np.sort(rng.integers(0, int(rng.integers(1, 6)), size=l)).tolist()
        pos = np.concatenate([np.sort(rng.integers(1, int(rng.integers(50, 5000)), size=chrom.count(c))) for c in sorted(set(chrom))]).tolist()
        w, s, m = int(rng.integers(1, 500)), int(rng.integers(1, 300)), int(rng.integers(1, 6))
        h, t = oracle.sliding_windows(chrom, pos, w, s, m)
        assert run(chrom, pos, w, s, m) == (h.tolist(), t.tolist())


def test_header_is_plain_c_and_links(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 (what cgo / bindgen / ctypes users see), and a C program
    linked against the library must resolve every entry point it names."""
    src = tmp_path / "abi.c"
    src.write_text('#include "poolgen_hip.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '    pg_ctx *c = 0;\n'
                   '    int rc = pg_create(&c, 0, 0);            /* no GPU here: must fail loudly, not fall back */\n'
                   '    printf("%d %s\\n", rc, pg_version());\n'
                   '    int64_t head[4], tail[4]; int32_t chr[4] = {0, 0, 0, 1}; uint64_t pos[4] = {1, 40, 200, 5};\n'
                   '    printf("%lld\\n", (long long)pg_host_sliding_windows(chr, pos, 4, 100, 50, 1, head, tail));\n'
                   '    if (c) pg_destroy(c);\n'
                   '    return 0;\n}\n')
    exe = tmp_path / "abi"
    libdir = ROOT / "poolgen_amd" / "csrc"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe),
                           "-L", str(libdir), "-lpoolgen_hip", f"-Wl,-rpath,{libdir}"])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    first, second = r.stdout.splitlines()[:2]
    if not torch.cuda.is_available():
        assert first.split()[0] != "0"                    # pg_create refused
    assert second == "3"                                   # [0..1], [2], [3]: what the oracle gives for these four loci
