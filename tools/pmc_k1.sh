#!/bin/bash
# K1/K23 instruction-mix counters: separate --pmc passes, kernel-trace only (dev tool, run under gpurun)
cd /tmp && export TMPDIR=/tmp
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}
OUT=$R/gpurun_out/pmc_k1
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o p$i -- python3 $R/tools/exp_iter.py 1000000 > $OUT/log$i.txt 2>&1
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -E "^nn_fast|^nn_tile|^accumulate_ell" -A28 $OUT/summary.txt | head -90
