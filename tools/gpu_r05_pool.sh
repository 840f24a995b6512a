#!/bin/bash
# round 5: the device pool — whole GPU suite, C++ API test, the cpp_api A/B (old = 02573b1) and the default bench line
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_pool; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
P=probabilistic_point_clouds_registration_amd
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/summary.txt
./$P/ppcr_cpp_api_test > $OUT/cpp_api_test.log 2>&1; echo "cpp_api_test rc=$?" >> $OUT/summary.txt
python - "$OUT" <<'PY'
import sys, numpy as np
from probabilistic_point_clouds_registration_amd import synth
src, tgt = synth.make_config(3)[:2]
np.ascontiguousarray(src[:, :3], dtype=np.float32).tofile(sys.argv[1] + "/src.f32")
np.ascontiguousarray(tgt[:, :3], dtype=np.float32).tofile(sys.argv[1] + "/tgt.f32")
PY
for rep in 1 2; do
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
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/cpp_api_test.log
grep -A1 "==" $OUT/ab.txt | grep -v "^--" | paste - - | cut -c1-330
head -8 $OUT/trace.txt | cut -c1-400
python - $OUT/bench_n1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('n1', round(d['value'],1), d['ms_per_step'], 'frac', r['frac'], 'conv', d['converged_inner']['it_per_s'], 'ttc', d['time_to_converge_ms']['value'], 'setup', d['setup_ms']['total'])
print(json.dumps(d['cpp_api'])[:1500])
PY
