#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_verlet3; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/tools/exp_verlet.py 1000000 > $OUT/exp_1m.txt 2>&1; echo "rocprof exp rc=$?" >> $OUT/summary.txt
python3 $R/tools/summarize_rocprof.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_summary.csv
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu > $OUT/parity.log 2>&1; echo "parity rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; cat $OUT/exp_1m.txt | cut -c1-250; cut -c1-200 $OUT/kernel_stats_summary.csv | head -30; grep -v "^  File" $OUT/parity.log | tail -12 | cut -c1-300
rm -rf $OUT/stats
