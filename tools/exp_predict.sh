#!/bin/bash
# prediction pass of the ridge path: time per launch, fraction of 8 TB/s, wall time of the job
set -e
mkdir -p gpurun_out
if [ "$1" = "tests" ]; then
timeout -k 10 500 python -m pytest tests/test_gpu_kinship_path.py tests/test_gpu_exact.py tests/test_gpu_cv.py -x -q -m gpu -k "ridge or penal or gp or cv or cross" > gpurun_out/predict_tests.log 2>&1 || { tail -30 gpurun_out/predict_tests.log; exit 1; }
tail -3 gpurun_out/predict_tests.log
fi
for n in 500 200 100; do
  echo "== n=$n"; timeout -k 10 200 python tools/bench_ridge.py $n 2000000 2 10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['predict_ms_total']/d['predict_launches'], d['predict_frac_of_8TBs'], d['wall_s'])"
done
