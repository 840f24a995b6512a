#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_verlet4; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/tools/exp_verlet.py 1000000 > $OUT/exp_1m.txt 2>&1; echo "rocprof exp rc=$?" >> $OUT/summary.txt
python3 $R/tools/summarize_rocprof.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_summary.csv
export PPCR_HIP_LIB=$R/probabilistic_point_clouds_registration_amd/_variants/libppcr_hip_scanorder.so
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o stats -- python3 $R/tools/exp_verlet.py 1000000 > $OUT/exp_1m_scanorder.txt 2>&1; echo "rocprof exp scanorder rc=$?" >> $OUT/summary.txt
python3 $R/tools/summarize_rocprof.py $(find $OUT/stats2 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_summary_scanorder.csv
unset PPCR_HIP_LIB
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu > $OUT/parity.log 2>&1; echo "parity rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; grep -v "^[WE]2026" $OUT/exp_1m.txt | cut -c1-250; cut -c1-200 $OUT/kernel_stats_summary.csv | head -8;  grep -v "^[WE]2026" $OUT/exp_1m_scanorder.txt | cut -c1-250; cut -c1-200 $OUT/kernel_stats_summary_scanorder.csv | head -8; grep -v "^  File" $OUT/parity.log | tail -12 | cut -c1-300
rm -rf $OUT/stats $OUT/stats2
