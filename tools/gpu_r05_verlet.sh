#!/bin/bash
# round 5: Verlet lists — first measurements, then the parity tests that move the source
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_verlet; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
timeout 600 python tools/exp_verlet.py 1000000 > $OUT/exp_1m.txt 2>&1; echo "exp 1M rc=$?" >> $OUT/summary.txt
timeout 600 python tools/exp_verlet.py 100000 > $OUT/exp_100k.txt 2>&1; echo "exp 100k rc=$?" >> $OUT/summary.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; cat $OUT/exp_1m.txt $OUT/exp_100k.txt; tail -25 $OUT/pytest.log
