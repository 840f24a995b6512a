#!/bin/bash
# round 5: the other configurations with Verlet lists on (default) and off
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/r05_configs; rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
for cfg in 2 5 3; do
  for v in 1 0; do
    timeout 900 python bench.py --config $cfg --opt verlet=$v --no-cpu-baseline --no-cpp-api > $OUT/cfg${cfg}_verlet$v.json 2> $OUT/cfg${cfg}_verlet$v.err
    python - $OUT/cfg${cfg}_verlet$v.json $cfg $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ss=d.get('steady_state',{})
    print('cfg',sys.argv[2],'verlet',sys.argv[3],'value',round(d['value'],1),'ms/step',round(d['ms_per_step'],4),'converged',(ss.get('converged') or {}).get('it_per_s'),'searched',(ss.get('window') or {}).get('searched_share'), 'conv_inner', (d.get('converged_inner') or {}).get('it_per_s'), 'ttc', (d.get('time_to_converge_ms') or {}).get('value'))
except Exception as e:
    print('cfg',sys.argv[2],'verlet',sys.argv[3],'ERROR',e)
PY
  done
done
