#!/bin/bash
# round 4: late kernarg reads (no SGPR spills in the folded K1) against the early-argument build, then the whole suite
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04l; rm -rf $OUT; mkdir -p $OUT; cd $R
V=$R/probabilistic_point_clouds_registration_amd/_variants
Q="python bench.py --no-extras --no-cpu-baseline"
for rep in 1 2 3; do
  $Q > $OUT/late_$rep.json 2>> $OUT/bench.err
  PPCR_HIP_LIB=$V/libppcr_hip_early.so $Q > $OUT/early_$rep.json 2>> $OUT/bench.err
done
timeout 1700 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/summary.txt
for f in late_1 early_1 late_2 early_2 late_3 early_3; do python - $OUT/$f.json <<'PY' >> $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'spread', round(d['windows']['spread'],4), 'k1_ms', r.get('avg_kernel_ms'), 'alone', (r.get('standalone') or {}).get('avg_kernel_ms'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $OUT/summary.txt; tail -4 $OUT/pytest_gpu.log
