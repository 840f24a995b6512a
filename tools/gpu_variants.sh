#!/bin/bash
# A/B of K1 translation-unit variants (tools/build_variant.py): the driver's bench command without the extras, two rounds
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/variants; rm -rf $OUT; mkdir -p $OUT; cd $R
V=$R/probabilistic_point_clouds_registration_amd/_variants
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-cpp-api"
for round in 1 2; do
  $B > $OUT/main_$round.json 2>> $OUT/err.txt
  for f in $V/libppcr_hip_*.so; do n=$(basename $f .so | sed 's/libppcr_hip_//'); PPCR_HIP_LIB=$f $B > $OUT/${n}_$round.json 2>> $OUT/err.txt; done
done
python - $OUT <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d['value'],1), d['ms_per_step'], 'k1', d['roofline'].get('avg_kernel_ms'))
    except Exception as e: print(os.path.basename(f), 'ERR', e)
PY
