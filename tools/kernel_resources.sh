#!/bin/bash
# Per-kernel register / LDS / spill figures of one translation unit, from the compiler's resource-usage remarks
# (no GPU needed).  usage: tools/kernel_resources.sh <file.hip> [extra hipcc flags, e.g. -DPPCR_TILE_M=10]
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I $R/include \
  -I $R/probabilistic_point_clouds_registration_amd/csrc -c "$@" $SRC -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines() if rows else []
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("ppcr::dev::", "")
    print("%-70s VGPR %3d AGPR %3d SGPR %3d spillV %2d spillS %3d scratch %5d occ %2d LDS %6d" % (
        n[:70], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", r.get("SGPRs", -1)), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
        r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))
'
