#!/bin/bash
# rocprofv3 kernel stats of bench.py --config <k> (no extras): usage: bash tools/gpu_prof_config.sh 8 10 ...
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/prof_config; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c$cfg -o stats -- python3 $R/bench.py --config $cfg --no-extras --no-cpu-baseline --no-cpp-api --no-profile > $OUT/c$cfg.log 2>&1
  python3 $R/tools/summarize_rocprof.py $(find $OUT/c$cfg -name "*kernel_stats.csv" | head -1) $OUT/c$cfg.csv
  echo "== config $cfg: $(tail -1 $OUT/c$cfg.log | cut -c1-140)"; head -9 $OUT/c$cfg.csv | cut -c1-150
  rm -rf $OUT/c$cfg
done
