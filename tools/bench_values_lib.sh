#!/bin/bash
# Dev aid (under gpurun): bench values of the configurations named, the tree's library against a variant library, alternating.
# usage: bench_values_lib.sh <variant .so> <cfg …>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; V=$1; shift
for rep in 1 2; do
  for lib in "" "$V"; do
    for cfg in "$@"; do
      PPCR_HIP_LIB=$lib python3 $R/bench.py --config $cfg --no-cpu-baseline --no-cpp-api --no-extras 2>/dev/null | tail -1 > /tmp/bv.json
      python3 -c "import json; d=json.load(open('/tmp/bv.json')); print('lib', '${lib:-tree}'.split('/')[-1], 'cfg', $cfg, round(d['value'],1), 'windows', [round(x) for x in d['windows']['it_per_s']])"
    done
  done
done
