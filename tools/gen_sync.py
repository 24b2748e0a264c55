#!/usr/bin/env python3
"""usage: gen_sync.py <out.sync> <out_phen.csv> <pools> <loci> -- synthetic sync text + phenotype file (for timing the CLI end to end)"""
import sys
import numpy as np
out, phen, n, L = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(1)
with open(out, "w") as f:
    for l0 in range(0, L, 20000):
        m = min(20000, L - l0)
        base = np.clip(rng.beta(0.5, 0.5, size=(m, 1)), 0.02, 0.98)
        d = rng.poisson(60, size=(m, n)) + 10
        a = rng.binomial(d, np.clip(base + rng.normal(0, 0.08, size=(m, n)), 0, 1))
        rows = []
        for l in range(m):
            rows.append("chr%d\t%d\tN\t" % (1 + (l0 + l) * 5 // L, l0 + l + 1) +
                        "\t".join("%d:%d:0:0:0:0" % (a[l, i], d[l, i] - a[l, i]) for i in range(n)))
        f.write("\n".join(rows) + "\n")
with open(phen, "w") as f:
    f.write("#pool,size,trait\n")
    y = rng.normal(size=n)
    for i in range(n):
        f.write("pool%d,20,%r\n" % (i, float(y[i])))
