#!/bin/bash
# round 4, after the final evidence: the set-up path per configuration, the scenes with one inner step, the closing check
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r04_close; rm -rf $OUT; mkdir -p $OUT; cd $R
for k in 3 2 8 9 10; do echo "config $k"; python tools/exp_setup.py $k -k; done > $OUT/setup_path.txt 2>&1
for c in 9 10; do python bench.py --config $c --inner-steps 1 --no-cpp-api --no-cpu-baseline > $OUT/bench_cfg${c}_inner1.json 2>> $OUT/bench.err; done
bash tools/gpu_check.sh > $OUT/check.log 2>&1
cp $R/gpurun_out/check/summary.txt $OUT/check_summary.txt
cat $OUT/setup_path.txt | grep -v "^    "; cat $OUT/check_summary.txt
