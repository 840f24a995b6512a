#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_verlet6; rm -rf "$OUT"; mkdir -p "$OUT"; cd $R
python3 tools/exp_verlet_bench.py 1000000 > $OUT/bench_1m.txt 2>&1
python3 tools/exp_verlet_bench.py 1000000 verlet_skin=100 > $OUT/bench_1m_skin100.txt 2>&1
python3 tools/exp_verlet_bench.py 100000 > $OUT/bench_100k.txt 2>&1
python3 tools/exp_verlet_dbg.py 30000 10 > $OUT/dbg.txt 2>&1
for f in bench_1m bench_1m_skin100 bench_100k; do echo "== $f"; cut -c1-330 $OUT/$f.txt; done; grep -v "^   row" $OUT/dbg.txt | cut -c1-200 | head -13
