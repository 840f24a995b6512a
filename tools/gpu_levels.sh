#!/bin/bash
# multi-level search: the scene tests and soaks, then configs 9 / 10 / 8, per-level counters
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/levels; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "cli_default_shape or two_pass or bit_identical or wave_kernel or row_per_wave or soak or dense or multi_level" > $OUT/pytest_levels.log 2>&1; echo "pytest levels rc=$?" >> $OUT/summary.txt
Q="python bench.py --no-extras --no-cpu-baseline"
for cfg in 9 10 8; do
  $Q --config $cfg > $OUT/cfg${cfg}.json 2>> $OUT/bench.err
  $Q --config $cfg --inner-steps 1 > $OUT/cfg${cfg}_inner1.json 2>> $OUT/bench.err
done
python tools/exp_levels.py > $OUT/levels.txt 2>&1
for f in cfg9 cfg9_inner1 cfg10 cfg10_inner1 cfg8 cfg8_inner1; do python - $OUT/$f.json <<'PY' >> $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'spread', round(d['windows']['spread'],4), {k: round(v*1e3,1) for k,v in d.get('kernels_ms_per_launch',{}).items()})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $OUT/summary.txt; tail -4 $OUT/pytest_levels.log; cat $OUT/levels.txt
