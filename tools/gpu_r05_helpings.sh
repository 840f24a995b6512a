#!/bin/bash
# round 5: wide one-pass K23 rows in two helpings (accumulate_ell_kernel / inner_steps_kernel, widths 16 / 20 / 32):
# the tests that compare device-paced and host-paced loops bit for bit, then configs 8 / 9 / 10 against the previous library
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_helpings; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_a.log 2>&1; echo "pytest pipeline+configs rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_a.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_pass or soak or align or inner or weights or moments or k23" > $OUT/pytest_b.log 2>&1; echo "pytest parity subset rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_b.log
OLD=$R/probabilistic_point_clouds_registration_amd/_variants/libppcr_hip_old.so
for round in 1 2; do
  for cfg in 8 10 9; do
    B="python bench.py --config $cfg --no-cpu-baseline --no-cpp-api --no-profile"
    $B > $OUT/new_c${cfg}_$round.json 2>> $OUT/err.txt
    PPCR_HIP_LIB=$OLD $B > $OUT/old_c${cfg}_$round.json 2>> $OUT/err.txt
  done
done
python - $OUT <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+'/*_c*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d['value'],1), round(d['ms_per_step'],4), 'ttc', round(d['time_to_converge_ms']['value'],3), 'cold', [round(x,3) for x in d['cold_ms_per_iteration']])
    except Exception as e: print(os.path.basename(f), 'ERR', e)
PY
