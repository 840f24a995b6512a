#!/bin/bash
# long seeded soaks (multi-level search, row-per-wave kernel, association, align) under seeds other than the suite's
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/soak; rm -rf $OUT; mkdir -p $OUT; cd $R
for seed in 11 303 9001; do
  PPCR_SOAK_SEED=$seed PPCR_SOAK_TRIALS=32 timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "soak" > $OUT/soak_$seed.log 2>&1; echo "soaks seed $seed trials 32 rc=$?" >> $OUT/summary.txt
done
cat $OUT/summary.txt; for seed in 11 303 9001; do tail -4 $OUT/soak_$seed.log; done
