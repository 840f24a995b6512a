#!/bin/bash
# round 5: double-buffered level feedback of the multi-level K1 — run-to-run partition, the scene tests, configs 9 / 10
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_levelfb; rm -rf $OUT; mkdir -p $OUT; cd $R
python tools/exp_level_partition.py > $OUT/partition.txt 2>&1; cat $OUT/partition.txt | tail -5
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_a.log 2>&1; echo "pytest configs rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/pytest_a.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level or multi or scene or wave_kernel or row_per_wave or two_pass" > $OUT/pytest_b.log 2>&1; echo "pytest parity subset rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/pytest_b.log
for cfg in 9 10; do for k in 1 2; do python bench.py --config $cfg --no-extras --no-cpu-baseline --no-cpp-api --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg', $cfg, round(d['value'],1))"; done; done
