#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point pg_ols_kinship (H2D of G, both passes, D2H of the
results).  Never the bench `value`; quoted in DESIGN.md section 6."""
import ctypes as C, json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poolgen_amd import Engine, synth

n, p = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
eng = Engine(0)
Gd = synth.genotype_matrix(p, n, "cuda")
Y = synth.phenotypes(Gd[:4096], n, k=1)
pinned = torch.empty((p, n), dtype=torch.float64, pin_memory=True)
pinned.copy_(Gd); torch.cuda.synchronize()
pageable = pinned.numpy().copy()
out = [np.empty((p, 1)) for _ in range(3)]
lib, ctx = eng._lib, eng._ctx
m = C.c_int()
for name, arr in (("pageable", pageable), ("pinned", pinned.numpy())):
    for rep in range(2):
        t0 = time.perf_counter()
        rc = lib.pg_ols_kinship(ctx, arr.ctypes.data, p, n, n, Y.ctypes.data, 1, 0.75, -1, C.byref(m), None,
                                out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data)
        dt = time.perf_counter() - t0
    assert rc == 0
    print(json.dumps({"host_buffer": name, "loci": p, "seconds": dt, "loci_per_s": p / dt,
                      "h2d_gbs_equiv": 8.0 * n * p / dt / 1e9, "m": m.value}))
