#!/bin/bash
# round 4, second GPU call: K23 XCD tile map + leaner inner steps (tests, stamps), K1 list-column permutation and unclamped
# list stores as A/B variants (bench + LDS bank-conflict counters), LDS out-of-range probe
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04b; rm -rf $OUT; mkdir -p $OUT; cd $R
V=$R/probabilistic_point_clouds_registration_amd/_variants
./tools/micro/lds_oob > $OUT/lds_oob.txt 2>&1; echo "lds_oob rc=$?" >> $OUT/summary.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/summary.txt
PPCR_HIP_LIB=$V/libppcr_hip_noclamp.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -m gpu > $OUT/pytest_noclamp.log 2>&1; echo "pytest noclamp rc=$?" >> $OUT/summary.txt
Q="python bench.py --no-extras --no-cpu-baseline"
for rep in 1 2; do
  $Q > $OUT/main_$rep.json 2>> $OUT/bench.err
  PPCR_HIP_LIB=$V/libppcr_hip_perm0.so $Q > $OUT/perm0_$rep.json 2>> $OUT/bench.err
  PPCR_HIP_LIB=$V/libppcr_hip_noclamp.so $Q > $OUT/noclamp_$rep.json 2>> $OUT/bench.err
done
python bench.py --no-extras --no-cpu-baseline --no-profile --inner-steps 100 > $OUT/inner100.json 2>> $OUT/bench.err
python tools/exp_inner_stamps.py > $OUT/inner_stamps.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in main perm0 noclamp; do
  L=""; [ $v != main ] && L=$V/libppcr_hip_$v.so
  PPCR_HIP_LIB=$L rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $OUT/pmc_$v -o p -- python3 $R/tools/exp_align.py 1000000 > $OUT/log_pmc_$v.txt 2>&1
  python3 $R/tools/pmc_summary.py $OUT/pmc_$v > $OUT/pmc_$v.txt 2>&1
  rm -rf $OUT/pmc_$v
done
cd $R
for f in main_1 perm0_1 noclamp_1 main_2 perm0_2 noclamp_2 inner100; do python - $OUT/$f.json <<'PY' >> $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'spread', round(d['windows']['spread'],4), 'k1_ms', r.get('avg_kernel_ms'), 'alone', (r.get('standalone') or {}).get('avg_kernel_ms'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $OUT/summary.txt; cat $OUT/lds_oob.txt; tail -3 $OUT/inner_stamps.txt; tail -4 $OUT/pytest_gpu.log; tail -4 $OUT/pytest_noclamp.log
for v in main perm0 noclamp; do echo $v; grep -A7 "^nn_fast_kernel<10, 16, 1728, false, 8>" $OUT/pmc_$v.txt; done
