#!/bin/bash
# the round's closing check: whole GPU suite, C++ API test, smoke, the driver's bench command
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/check; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/summary.txt
./probabilistic_point_clouds_registration_amd/ppcr_cpp_api_test > $OUT/cpp_api_test.log 2>&1; echo "cpp_api_test rc=$?" >> $OUT/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?" >> $OUT/summary.txt
python - $OUT/bench_n1.json <<'PY' >> $OUT/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('n1', round(d['value'],1), d['ms_per_step'], 'frac', r['frac'], 'conv', d['converged_inner']['it_per_s'], 'ttc', d['time_to_converge_ms']['value'], 'setup', d['setup_ms']['total'])
PY
cat $OUT/summary.txt; tail -4 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
