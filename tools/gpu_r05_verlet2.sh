#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_verlet2; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
timeout 600 python tools/exp_verlet.py 1000000 > $OUT/exp_1m.txt 2>&1; echo "exp 1M rc=$?" >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "randomised_association_soak" > $OUT/soak.log 2>&1; echo "soak rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; cat $OUT/exp_1m.txt; grep -v "^  File" $OUT/soak.log | tail -30 | cut -c1-300
