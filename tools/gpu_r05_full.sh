#!/bin/bash
# round 5: whole GPU suite, C++ API test, smoke, default bench line
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_full; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
P=probabilistic_point_clouds_registration_amd
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/summary.txt
./$P/ppcr_cpp_api_test > $OUT/cpp_api_test.log 2>&1; echo "cpp_api_test rc=$?" >> $OUT/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; grep -E "passed|failed|FAILED|Error" $OUT/pytest_gpu.log | tail -15; tail -1 $OUT/cpp_api_test.log; tail -3 $OUT/bench_n1.err
python - $OUT/bench_n1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('n1', round(d['value'],1), d['ms_per_step'], 'frac', r['frac'], 'conv', d['converged_inner']['it_per_s'], 'ttc', d['time_to_converge_ms']['value'], 'setup', d['setup_ms']['total'])
print(json.dumps(d['roofline'])[:600]); print(json.dumps(d['cpp_api'])[:900]); print('parity', json.dumps(d.get('parity'))[:300])
PY
