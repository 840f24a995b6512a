#!/bin/bash
# round 4: the multi-level soak (default seed + three others), the other soaks under two other seeds
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04k; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_level_soak" > $OUT/soak_default.log 2>&1; echo "soak default rc=$?" >> $OUT/summary.txt
for seed in 11 2024 987654; do
  PPCR_SOAK_SEED=$seed timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "soak" > $OUT/soak_$seed.log 2>&1; echo "soaks seed $seed rc=$?" >> $OUT/summary.txt
done
cat $OUT/summary.txt; tail -15 $OUT/soak_default.log; for seed in 11 2024 987654; do tail -3 $OUT/soak_$seed.log; done
