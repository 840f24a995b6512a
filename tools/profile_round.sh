#!/bin/bash
# Round profile (run under gpurun): rocprofv3 kernel stats of the default bench command, then separate --pmc
# passes for the HBM traffic of the kernels (FETCH_SIZE / WRITE_SIZE, as MI355X_MICROARCH.md prescribes).
# usage: bash tools/profile_round.sh <tag>      outputs under gpurun_out/prof_<tag>/
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/pmc_$set -o pmc -- python3 $R/tools/exp_iter.py 1000000 > $OUT/log_$set.txt 2>&1
done
python3 $R/tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_hbm_traffic.txt 2>&1
python3 $R/tools/make_k1_traffic.py $OUT/pmc_hbm_traffic.txt $OUT/k1_traffic.json $TAG > /dev/null 2>&1
python3 $R/tools/summarize_rocprof.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_summary.csv
cd $R && python3 bench.py > $OUT/bench.json 2>$OUT/bench.err
tail -1 $OUT/bench_under_rocprof.log | cut -c1-160
head -12 $OUT/kernel_stats_summary.csv
grep -E "^nn_fast|^nn_tile|^accumulate_ell" -A3 $OUT/pmc_hbm_traffic.txt
