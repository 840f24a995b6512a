#!/bin/bash
# round 4, first GPU call: new tests + host-budget measurement (config 5 under taskset) + tail experiment + inner-step stamps
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04a; rm -rf $OUT; mkdir -p $OUT; cd $R
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1
./probabilistic_point_clouds_registration_amd/ppcr_cpp_api_test > $OUT/cpp_api_test.log 2>&1; echo "cpp_api_test rc=$?" >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "batch_entry_points" > $OUT/pytest_cfg5.log 2>&1; echo "pytest cfg5 rc=$?" >> $OUT/summary.txt
B="python bench.py --config 5 --lanes 4 --no-extras --no-cpu-baseline --no-verify --no-profile"
$B > $OUT/cfg5_free.json 2> $OUT/cfg5_free.err
taskset -c 0 $B > $OUT/cfg5_taskset1.json 2> $OUT/cfg5_taskset1.err
taskset -c 0,1 $B > $OUT/cfg5_taskset2.json 2> $OUT/cfg5_taskset2.err
Q="python bench.py --no-extras --no-cpu-baseline"
$Q > $OUT/n1_1000000.json 2> $OUT/n1.err
$Q --n 983040 > $OUT/n1_983040.json 2>> $OUT/n1.err
for s in 1 2 3; do python bench.py --no-extras --no-cpu-baseline --no-profile --inner-steps 100 --opt inner_dev_steps=$s > $OUT/inner_dev$s.json 2>> $OUT/n1.err; done
python tools/exp_inner_stamps.py > $OUT/inner_stamps.txt 2>&1
for f in cfg5_free cfg5_taskset1 cfg5_taskset2 n1_1000000 n1_983040 inner_dev1 inner_dev2 inner_dev3; do python - $OUT/$f.json <<'PY' >> $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'spread', round(d['windows']['spread'],4), 'k1_ms', r.get('avg_kernel_ms'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $OUT/summary.txt; tail -3 $OUT/inner_stamps.txt; tail -3 $OUT/cpp_api_test.log; tail -5 $OUT/pytest_cfg5.log
