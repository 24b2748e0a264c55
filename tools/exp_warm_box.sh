#!/bin/bash
# GPU box: does the bench line depend on what the box did before?  A: fresh; B: after five more bench runs back to back;
# C: right after a rocprofv3 --pmc pass; D: after two idle minutes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
line() { python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1: step %.3f ms  kinship %.3f  sweep %.3f  m8 sweep %.3f' % (d['ms_per_step'], d['kernels']['k_kinship_syrk']['avg_ms'], d['roofline_sweep']['two_pass']['avg_ms'], d['roofline_sweep']['m8']['avg_ms']))"; }
line A
for i in 1 2 3 4; do python3 bench.py --no-cpu-baseline --no-secondary > /dev/null 2>&1; done
line B
rm -rf gpurun_out/warm_pmc; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/warm_pmc -- python3 bench.py --steps 3 --warmup 1 --sweep-steps 2 --no-cpu-baseline > /dev/null 2>&1; rm -rf gpurun_out/warm_pmc
line C
sleep 120
line D
