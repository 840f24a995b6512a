#!/bin/bash
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/setup; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/summary.txt
for k in 3 9 10; do echo "config $k" >> $OUT/summary.txt; python tools/exp_setup.py $k >> $OUT/summary.txt 2>&1; done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpp-api > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python - $OUT/bench_n1.json <<'PY' >> $OUT/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('n1', round(d['value'],1), d['ms_per_step'])
print('setup', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['setup_ms'].items() if k!='note'}); print('cold', [round(x,3) for x in d['cold_ms_per_iteration']]); print('ttc', d['time_to_converge_ms']['value'])
PY
cat $OUT/summary.txt; tail -3 $OUT/pytest.log
