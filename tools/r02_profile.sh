#!/bin/bash
# Round-2 evidence for the bench line, all from ONE command line (python3 bench.py [flags]) on one MI355X box:
#   1. the bench JSON itself                                   -> gpurun_out/r02_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command    -> gpurun_out/r02_kernel_stats.csv (our kernels' rows)
#   3. rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy), one counter group per pass, --kernel-trace only
#                                                             -> gpurun_out/r02_pmc_summary.txt, gpurun_out/r02_pmc_traffic.json
# Copy the four files into profiles/ afterwards (profiles/pmc_traffic.json is what bench.py reads for `traffic`).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench rc=$?"
rm -rf gpurun_out/r02_prof && mkdir -p gpurun_out/r02_prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof -- python3 bench.py --no-cpu-baseline > gpurun_out/r02_prof.log 2>&1; echo "rocprof rc=$?"
f=$(find gpurun_out/r02_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/r02_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(rows[0])
for r in rows[1:]:
    if "k_" in r[0] and "at::" not in r[0]:
        r[0] = r[0].replace("(anonymous namespace)::", "")
        w.writerow(r)
PY
cat gpurun_out/r02_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r02_prof
: > gpurun_out/r02_pmc_summary.txt
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/r02_pmc_$i && mkdir -p gpurun_out/r02_pmc_$i
  timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/r02_pmc_$i -- python3 bench.py --steps 3 --warmup 1 --sweep-steps 2 --no-cpu-baseline > gpurun_out/r02_pmc_$i.log 2>&1
  echo "pmc pass $i ($ctrs) rc=$?"
  f=$(find gpurun_out/r02_pmc_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> gpurun_out/r02_pmc_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    if not name.startswith("k_"): continue
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    for c, v in d.items():
        print(f"{k:44s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
  rm -rf gpurun_out/r02_pmc_$i
done
python3 - <<'PY'
import json, re
vals = {}
for line in open("gpurun_out/r02_pmc_summary.txt"):
    m = re.match(r"(.{44})\s(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
    if m: vals[(m.group(1).strip(), m.group(2))] = float(m.group(4))
def hbm(k):  # gfx950: FETCH_SIZE (KiB) counts half of a wide streaming read (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact
    f, w = vals.get((k, "FETCH_SIZE")), vals.get((k, "WRITE_SIZE"))
    return None if f is None or w is None else int((2 * f + w) * 1024)
out = {"_comment": "HBM bytes per launch from rocprofv3 --pmc passes of `python3 bench.py --steps 3 --warmup 1 --sweep-steps 2 --no-cpu-baseline` "
                   "(tools/r02_profile.sh; per-kernel means in r02_pmc_summary.txt): (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the factor 2 being the "
                   "gfx950 wide-read correction of MI355X_MICROARCH.md",
       "source": "profiles/r02_pmc_summary.txt", "workload": "200x10000000",
       "kinship_hbm_bytes_per_launch": hbm("k_kinship_syrk<true, true, 3>"),
       "kinship_two_pass_hbm_bytes_per_launch": hbm("k_kinship_syrk<false, true, 3>"),
       # (one kernel serves both legs since the matrix-core sweep: the per-kernel mean is over the launches of both)
       "sweep_two_pass_hbm_bytes_per_launch": hbm("k_ols_sweep_mfma<5, 3, 1, 0>"),
       "sweep_m8_hbm_bytes_per_launch": hbm("k_ols_sweep_mfma<5, 3, 1, 0>")}
mf, ga = vals.get(("k_kinship_syrk<true, true, 3>", "SQ_VALU_MFMA_BUSY_CYCLES")), vals.get(("k_kinship_syrk<true, true, 3>", "GRBM_GUI_ACTIVE"))
if mf and ga: out["kinship_mfma_busy_frac"] = mf / 1024.0 / (ga / 8.0)
json.dump(out, open("gpurun_out/r02_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r02_pmc_summary.txt
