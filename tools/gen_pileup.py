#!/usr/bin/env python3
"""usage: gen_pileup.py <out.pileup> <out_phen.csv> <pools> <loci> -- synthetic mpileup text + phenotype file (timing only)"""
import random, sys
out, phen, n, L = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rng = random.Random(1)
alphabet = "....,,,,TtTt"
with open(out, "w") as f:
    for l in range(L):
        parts = ["chr%d" % (1 + l * 3 // L), str(l + 1), rng.choice("ACG")]
        for _ in range(n):
            cov = rng.randint(20, 60)
            parts += [str(cov), "".join(rng.choices(alphabet, k=cov)), "J" * cov]
        f.write("\t".join(parts) + "\n")
with open(phen, "w") as f:
    f.write("#pool,size,trait\n")
    for i in range(n):
        f.write("pool%d,20,%r\n" % (i, rng.gauss(0, 1)))
