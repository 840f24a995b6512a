#!/bin/bash
# nn_wide_kernel knobs (loads in flight, list entries per wave, workgroups per CU): configs 9 / 10 / 8 with each variant
# built by tools/build_variant.py --width 20, two rounds
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/wide_variants; rm -rf $OUT; mkdir -p $OUT; cd $R
V=$R/probabilistic_point_clouds_registration_amd/_variants
for round in 1 2; do
  for cfg in 10 9 8; do
    B="python bench.py --config $cfg --no-extras --no-cpu-baseline --no-cpp-api --no-profile"
    $B > $OUT/main_c${cfg}_$round.json 2>> $OUT/err.txt
    for f in $V/libppcr_hip_*.so; do n=$(basename $f .so | sed 's/libppcr_hip_//'); PPCR_HIP_LIB=$f $B > $OUT/${n}_c${cfg}_$round.json 2>> $OUT/err.txt; done
  done
done
python - $OUT <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d['value'],1), round(d['ms_per_step'],4))
    except Exception as e: print(os.path.basename(f), 'ERR', e)
PY
