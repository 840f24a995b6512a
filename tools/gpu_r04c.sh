#!/bin/bash
# round 4, third GPU call: all tests at the new default (unclamped list stores, list-column deal, all-halves K1 for small
# clouds, native RCCL gather), small-cloud A/B (config 2: 100k), config 5 with halves, non-uniform configs 9 / 10
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1700 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/summary.txt
./probabilistic_point_clouds_registration_amd/ppcr_cpp_api_test > $OUT/cpp_api_test.log 2>&1; echo "cpp_api_test rc=$?" >> $OUT/summary.txt
Q="python bench.py --no-extras --no-cpu-baseline"
for rep in 1 2; do
  $Q --config 2 > $OUT/cfg2_auto_$rep.json 2>> $OUT/bench.err
  $Q --config 2 --opt k1_halves=0 > $OUT/cfg2_whole_$rep.json 2>> $OUT/bench.err
done
$Q --config 2 --n 50000 > $OUT/n50k_auto.json 2>> $OUT/bench.err
$Q --config 2 --n 50000 --opt k1_halves=0 > $OUT/n50k_whole.json 2>> $OUT/bench.err
$Q --config 2 --n 160000 > $OUT/n160k_auto.json 2>> $OUT/bench.err
$Q --config 2 --n 160000 --opt k1_halves=0 > $OUT/n160k_whole.json 2>> $OUT/bench.err
$Q --config 2 --n 250000 --opt k1_halves=1 > $OUT/n250k_halves.json 2>> $OUT/bench.err
$Q --config 2 --n 250000 > $OUT/n250k_whole.json 2>> $OUT/bench.err
$Q --config 5 --lanes 4 --no-verify --no-profile > $OUT/cfg5_whole.json 2>> $OUT/bench.err
$Q --config 5 --lanes 4 --no-verify --no-profile --opt k1_halves=1 > $OUT/cfg5_halves.json 2>> $OUT/bench.err
$Q > $OUT/n1.json 2>> $OUT/bench.err
$Q --config 9 > $OUT/cfg9.json 2>> $OUT/bench.err
$Q --config 10 > $OUT/cfg10.json 2>> $OUT/bench.err
$Q --config 8 > $OUT/cfg8.json 2>> $OUT/bench.err
for f in cfg2_auto_1 cfg2_whole_1 cfg2_auto_2 cfg2_whole_2 n50k_auto n50k_whole n160k_auto n160k_whole n250k_halves n250k_whole cfg5_whole cfg5_halves n1 cfg9 cfg10 cfg8; do python - $OUT/$f.json <<'PY' >> $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'spread', round(d['windows']['spread'],4), 'k1_ms', r.get('avg_kernel_ms'), {k: round(v*1e3,1) for k,v in d.get('kernels_ms_per_launch',{}).items()})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $OUT/summary.txt; tail -5 $OUT/pytest_gpu.log; tail -2 $OUT/cpp_api_test.log; tail -5 $OUT/bench.err
