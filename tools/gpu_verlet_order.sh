#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_verlet8; rm -rf "$OUT"; mkdir -p "$OUT"; cd $R
python3 tools/exp_verlet_dbg.py 30000 10 > $OUT/dbg.txt 2>&1
grep -v "^   row" $OUT/dbg.txt | cut -c1-110 | head -13
cd /tmp && export TMPDIR=/tmp
for ord in 1 0; do
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$ord -o tr -- python3 $R/tools/exp_verlet_bench.py 1000000 verlet_order=$ord > $OUT/bench_1m_order$ord.txt 2>&1
python3 - $OUT/tr$ord <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fast = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "nn_fast_kernel" in r["Kernel_Name"]]
print("last 25 nn_fast_kernel launches (us):", " ".join(f"{d:.0f}" for n, d in fast[-25:]), " sum of the last 20:", round(sum(d for n, d in fast[-20:])))
PY
grep -v "^[WE]2026" $OUT/bench_1m_order$ord.txt | grep "verlet=1" -A1 | cut -c1-300
rm -rf $OUT/tr$ord
done
cd $R; python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "verlet or soak or temporal or batch" 2>&1 | tail -4
