#!/bin/bash
# Copy what tools/gpu_r06_evidence.sh left under gpurun_out/prof_<tag>/ into profiles/ (tracked), named per round.
# usage: tools/collect_profiles.sh <tag> <round prefix, e.g. r06>
set -eu; R=$(cd "$(dirname "$0")/.." && pwd); S=$R/gpurun_out/prof_$1; P=$2
cp $S/bench.json $R/profiles/${P}_bench_n1.json
for c in 2 4 5 8 9 10; do [ -s $S/bench_cfg$c.json ] && cp $S/bench_cfg$c.json $R/profiles/${P}_bench_cfg$c.json; done
cp $S/kernel_stats_summary.csv $R/profiles/${P}_kernel_stats.csv
cp $S/kernel_stats_exp_align.csv $R/profiles/${P}_kernel_stats_exp_align.csv
for c in 2 8 10; do [ -s $S/kernel_stats_config$c.csv ] && cp $S/kernel_stats_config$c.csv $R/profiles/${P}_kernel_stats_config$c.csv; done
cp $S/pmc_hbm_traffic.txt $R/profiles/${P}_pmc_hbm_traffic.txt
[ -s $S/pmc_instruction_mix.txt ] && cp $S/pmc_instruction_mix.txt $R/profiles/${P}_pmc_instruction_mix.txt
python3 $R/tools/make_k1_traffic.py $R/profiles/${P}_pmc_hbm_traffic.txt $R/profiles/k1_traffic.json $P $R/profiles/${P}_pmc_instruction_mix.txt > /dev/null
ls -la $R/profiles/${P}_* | wc -l
