#!/bin/bash
# round 4: the tests added after the final evidence run (sequence numbers, wider LDS probe), set-up cost after the occupancy cache
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04j; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_bench_contract.py -x -q -m gpu -k "sequence_numbers or lds_stores or bench_line or run_ahead or paces" > $OUT/pytest_new.log 2>&1; echo "pytest new rc=$?" >> $OUT/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python - $OUT/bench_n1.json <<'PY' >> $OUT/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('n1', round(d['value'],1), d['ms_per_step'], {k:r.get(k) for k in ('kernel','frac','measured_traffic_frac','valu_issue_frac','valu_per_wave','lds_bank_conflict_share')})
print('setup', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['setup_ms'].items() if k!='note'}); print('cold', d['cold_ms_per_iteration']); print('ttc', d['time_to_converge_ms']['value'], 'conv', d['converged_inner']['it_per_s'])
PY
cat $OUT/summary.txt; tail -5 $OUT/pytest_new.log
