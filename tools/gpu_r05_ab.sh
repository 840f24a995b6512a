#!/bin/bash
# round 5: where does the fixed cost of align() on a fresh handle go?  Old (02573b1) and current build side by side on one box.
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_ab; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
P=probabilistic_point_clouds_registration_amd
python - "$OUT" <<'PY'
import sys, numpy as np
from probabilistic_point_clouds_registration_amd import synth
src, tgt = synth.make_config(3)[:2]
np.ascontiguousarray(src[:, :3], dtype=np.float32).tofile(sys.argv[1] + "/src.f32")
np.ascontiguousarray(tgt[:, :3], dtype=np.float32).tofile(sys.argv[1] + "/tgt.f32")
PY
for rep in 1 2 3; do
  for v in new old; do
    exe=$P/ppcr_cpp_api_test; [ $v = old ] && exe=$P/_variants/old_02573b1/ppcr_cpp_api_test
    for inner in 1 100; do
      echo "== $v inner=$inner rep=$rep" >> $OUT/ab.txt
      $exe --bench $OUT/src.f32 $OUT/tgt.f32 1.0 10 5.0 5 20 $inner 7 >> $OUT/ab.txt 2>&1
    done
  done
done
PPCR_TRACE=1 $P/ppcr_cpp_api_test --bench $OUT/src.f32 $OUT/tgt.f32 1.0 10 5.0 5 20 1 7 > $OUT/trace.txt 2>&1
rm -f $OUT/src.f32 $OUT/tgt.f32
grep -A1 "==" $OUT/ab.txt | python -c "
import sys,json
lab=None
for ln in sys.stdin:
    if ln.startswith('=='): lab=ln.strip()
    elif ln.startswith('{'):
        d=json.loads(ln); print(lab, 'steady', round(d['steady_it_per_s']), round(d['steady_min']), round(d['steady_max']), 'whole', round(d['whole_align_it_per_s']))
"
head -30 $OUT/trace.txt | cut -c1-600
