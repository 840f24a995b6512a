#!/bin/bash
# round 4, final evidence: rocprofv3 kernel stats + PMC passes (tools/profile_round.sh), the driver's bench command, the
# other configs' bench lines, the host-budget runs, the whole GPU suite
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r04_final; rm -rf $OUT; mkdir -p $OUT; cd $R
bash tools/profile_round.sh r04_final > $OUT/profile_round.log 2>&1
cp $R/gpurun_out/prof_r04_final/*.csv $R/gpurun_out/prof_r04_final/*.txt $R/gpurun_out/prof_r04_final/*.json $OUT/ 2>/dev/null
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_driver_cmd.json 2> $OUT/bench_n1.err
Q="python bench.py"
for cfg in 2 4 6 7; do $Q --config $cfg --no-cpp-api > $OUT/bench_cfg$cfg.json 2>> $OUT/bench.err; done
for cfg in 8 9 10; do $Q --config $cfg --no-cpp-api > $OUT/bench_cfg$cfg.json 2>> $OUT/bench.err; done
$Q --config 5 --lanes 4 > $OUT/bench_cfg5.json 2>> $OUT/bench.err
B="python bench.py --config 5 --lanes 4 --no-extras --no-cpu-baseline --no-verify --no-profile"
taskset -c 0 $B > $OUT/bench_cfg5_taskset1.json 2>> $OUT/bench.err
taskset -c 0,1 $B > $OUT/bench_cfg5_taskset2.json 2>> $OUT/bench.err
python tools/exp_inner_stamps.py > $OUT/inner_stamps.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/summary.txt
for f in bench_n1_driver_cmd bench_cfg2 bench_cfg4 bench_cfg6 bench_cfg7 bench_cfg8 bench_cfg9 bench_cfg10 bench_cfg5 bench_cfg5_taskset1 bench_cfg5_taskset2; do python - $OUT/$f.json <<'PY' >> $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'spread', round(d['windows']['spread'],4), 'k1_ms', r.get('avg_kernel_ms'), 'frac', r.get('frac'), 'conv', (d.get('converged_inner') or {}).get('it_per_s'), 'ttc', (d.get('time_to_converge_ms') or {}).get('value'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $OUT/summary.txt; tail -3 $OUT/inner_stamps.txt; tail -4 $OUT/pytest_gpu.log; head -14 $OUT/kernel_stats_summary.csv
