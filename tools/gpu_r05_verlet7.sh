#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_verlet7; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o tr -- python3 $R/tools/exp_verlet_bench.py 1000000 > $OUT/bench_1m.txt 2>&1
python3 - $OUT <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "nn_fast_kernel" in r["Kernel_Name"] or "cleanup" in r["Kernel_Name"]]
# the last 25-iteration align of the verlet=1 context: print the fast kernel's durations in launch order
fast = [(n, d) for n, d in seq if "nn_fast_kernel" in n]
print("last 60 nn_fast_kernel launches (us):")
print(" ".join(f"{d:.0f}{'v' if 'true>' in n and n.rstrip('>').endswith('true') else ''}" for n, d in fast[-60:]))
PY
grep -v "^[WE]2026" $OUT/bench_1m.txt | cut -c1-300
rm -rf $OUT/tr
