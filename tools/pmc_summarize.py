"""Summarise a rocprofv3 --pmc CSV: per kernel (ours only), mean counter value per dispatch."""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_kinship" in k or "k_ols" in k or "k_locus" in k or "k_gp" in k:
            short = "k_" + k.split("k_", 1)[1].split("(")[0][:36]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    for c, v in cs.items():
        print(f"{k:40s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
