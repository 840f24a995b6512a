#!/bin/bash
# round 4: everything at the current state: whole GPU suite, C++ API test, the headline, the non-uniform configs
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04i; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1700 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/summary.txt
./probabilistic_point_clouds_registration_amd/ppcr_cpp_api_test > $OUT/cpp_api_test.log 2>&1; echo "cpp_api_test rc=$?" >> $OUT/summary.txt
Q="python bench.py --no-extras --no-cpu-baseline"
$Q > $OUT/n1.json 2>> $OUT/bench.err
for cfg in 9 10; do
  $Q --config $cfg > $OUT/cfg${cfg}.json 2>> $OUT/bench.err
  $Q --config $cfg --inner-steps 1 > $OUT/cfg${cfg}_inner1.json 2>> $OUT/bench.err
done
for f in n1 cfg9 cfg9_inner1 cfg10 cfg10_inner1; do python - $OUT/$f.json <<'PY' >> $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'spread', round(d['windows']['spread'],4), {k: round(v*1e3,1) for k,v in d.get('kernels_ms_per_launch',{}).items()})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $OUT/summary.txt; tail -6 $OUT/pytest_gpu.log; tail -2 $OUT/cpp_api_test.log
