#!/bin/bash
# rocprofv3 kernel stats of tools/exp_iter.py (dev tool, run under gpurun): usage: bash tools/prof_iter.sh <tag> [exp_iter args]
TAG=${1:-x}; shift
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/tools/exp_iter.py "$@" > $OUT/log.txt 2>&1
python3 $R/tools/summarize_rocprof.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_summary.csv
head -14 $OUT/kernel_stats_summary.csv
tail -6 $OUT/log.txt
