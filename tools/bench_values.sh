#!/bin/bash
# Dev aid (under gpurun): the bench line's value for each configuration named, twice over.  usage: bench_values.sh 8 9 10
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
  for cfg in "$@"; do
    python3 $R/bench.py --config $cfg --no-cpu-baseline --no-cpp-api --no-extras 2>/dev/null | tail -1 > /tmp/bv.json
    python3 -c "import json; d=json.load(open('/tmp/bv.json')); print('cfg', $cfg, round(d['value'],1), d['unit'])"
  done
done
