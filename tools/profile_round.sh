#!/bin/bash
# Round profile (run under gpurun): rocprofv3 kernel stats of the default bench command, then separate --pmc
# passes (HBM traffic: FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md prescribes; instruction mix) over
# tools/exp_align.py, which launches the kernels exactly as the bench's timed windows do (ppcr_align, K23 folded
# into K1) and, behind them, the stand-alone forms.
# usage: bash tools/profile_round.sh <tag> [nopmcmix] [dof]    outputs under gpurun_out/prof_<tag>/   (dof: exp_align.py's model, default 5)
TAG=${1:-rXX}
set -u; R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/bench.py --no-cpu-baseline --no-cpp-api > $OUT/bench_under_rocprof.log 2>&1
python3 $R/tools/summarize_rocprof.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_summary.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_align -o stats -- python3 $R/tools/exp_align.py 1000000 dof=${3:-5} > $OUT/exp_align.log 2>&1
python3 $R/tools/summarize_rocprof.py $(find $OUT/stats_align -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_exp_align.csv
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/pmc_$set -o pmc -- python3 $R/tools/exp_align.py 1000000 dof=${3:-5} > $OUT/log_$set.txt 2>&1
done
python3 $R/tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --split "nn_fast_kernel<10, 24, 1920" 90000 > $OUT/pmc_hbm_traffic.txt 2>&1
if [ "${2:-}" != "nopmcmix" ]; then
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
             "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/mix$i -o p$i -- python3 $R/tools/exp_align.py 1000000 dof=${3:-5} > $OUT/log_mix$i.txt 2>&1
  done
  python3 $R/tools/pmc_summary.py $OUT/mix1 $OUT/mix2 $OUT/mix3 $OUT/mix4 > $OUT/pmc_instruction_mix.txt 2>&1
fi
MIX=""; [ -s $OUT/pmc_instruction_mix.txt ] && MIX=$OUT/pmc_instruction_mix.txt
python3 $R/tools/make_k1_traffic.py $OUT/pmc_hbm_traffic.txt $OUT/k1_traffic.json $TAG $MIX > /dev/null 2>&1
cd $R && python3 bench.py > $OUT/bench.json 2>$OUT/bench.err
tail -1 $OUT/bench_under_rocprof.log | cut -c1-160
head -12 $OUT/kernel_stats_summary.csv
head -12 $OUT/kernel_stats_exp_align.csv
grep -E "^nn_fast|^nn_tile|^accumulate_ell|^inner_steps" -A3 $OUT/pmc_hbm_traffic.txt
