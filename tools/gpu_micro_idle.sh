#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}; OUT=$R/gpurun_out/micro_idle; rm -rf "$OUT"; mkdir -p "$OUT"; cd $R/tools/micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 idle_launch.hip -o /tmp/idle_launch 2> $OUT/build.log && /tmp/idle_launch > $OUT/idle_launch.txt 2>&1
cat $OUT/build.log | head; cat $OUT/idle_launch.txt
