#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_verlet5; rm -rf "$OUT"; mkdir -p "$OUT"; cd $R
python3 tools/exp_verlet_dbg.py 30000 10 > $OUT/dbg.txt 2>&1
python3 tools/exp_verlet_dbg.py 30000 10 1 verlet_skin=2000 > $OUT/dbg_skin2000.txt 2>&1
python3 tools/exp_verlet_dbg.py 30000 5 30 > $OUT/dbg_m5_inner.txt 2>&1
for f in dbg dbg_skin2000 dbg_m5_inner; do echo "== $f"; grep -v "^   row" $OUT/$f.txt | cut -c1-220 | head -14; done
