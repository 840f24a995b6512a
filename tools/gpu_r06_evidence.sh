#!/bin/bash
# Round 6 evidence at one commit (run under gpurun): the round profile (tools/profile_round.sh), then the bench line of every
# other configuration — BASELINE configs[1] (100k), configs[3] (Gaussian 1M), configs[4] (64 x 250k on this GPU) and the
# command line's default shapes (--config 8 / 9 / 10) — and their rocprofv3 kernel stats.
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; TAG=${1:-r06}
bash $R/tools/profile_round.sh $TAG > $R/gpurun_out/profile_round_$TAG.log 2>&1
OUT=$R/gpurun_out/prof_$TAG; cd $R
for cfg in 2 4 5 8 9 10; do
  timeout 900 python bench.py --config $cfg --no-cpu-baseline --no-cpp-api > $OUT/bench_cfg$cfg.json 2> $OUT/bench_cfg$cfg.err
done
cd /tmp && export TMPDIR=/tmp
for cfg in 2 8 10; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$cfg -o stats -- python3 $R/bench.py --config $cfg --no-cpu-baseline --no-cpp-api --no-extras > $OUT/bench_cfg${cfg}_under_rocprof.log 2>&1
  python3 $R/tools/summarize_rocprof.py $(find $OUT/stats_cfg$cfg -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_config$cfg.csv
done
rm -rf $OUT/stats $OUT/stats_align $OUT/stats_cfg* $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/mix1 $OUT/mix2 $OUT/mix3 $OUT/mix4
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), "frac", d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
