#!/bin/bash
# round 4: multi-level grids with the 7/8-quantile level choice and overflow feedback: configs 8 / 9 / 10 with and without
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04e; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "cli_default_shape or two_pass or bit_identical or wave_kernel or row_per_wave or soak or dense" > $OUT/pytest_levels.log 2>&1; echo "pytest levels rc=$?" >> $OUT/summary.txt
Q="python bench.py --no-extras --no-cpu-baseline"
for cfg in 9 10 8; do
  $Q --config $cfg > $OUT/cfg${cfg}_levels.json 2>> $OUT/bench.err
  $Q --config $cfg --opt levels=0 > $OUT/cfg${cfg}_single.json 2>> $OUT/bench.err
done
python tools/exp_cli_shape.py > $OUT/cli_shape.txt 2>&1
for f in cfg9_levels cfg9_single cfg10_levels cfg10_single cfg8_levels cfg8_single; do python - $OUT/$f.json <<'PY' >> $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'spread', round(d['windows']['spread'],4), {k: round(v*1e3,1) for k,v in d.get('kernels_ms_per_launch',{}).items()})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $OUT/summary.txt; tail -4 $OUT/pytest_levels.log; grep -v "^$" $OUT/cli_shape.txt | head -30
