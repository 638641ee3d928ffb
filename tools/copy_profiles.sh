#!/bin/bash
# copy what tools/collect_profiles.sh left in gpurun_out/<tag>/ into profiles/<prefix>_* (the files the judge reads)
# usage: bash tools/copy_profiles.sh r04prof4 r04
P=gpurun_out/$1; R=profiles/$2
cp $P/bench_default.json ${R}_bench_default.json; cp $P/bench_under_rocprof.json ${R}_bench_under_rocprof.json
cp $P/stats_cfg2/t_kernel_stats.csv ${R}_kernel_stats_bench_steps3.csv
cp $P/this_traffic.json ${R}_traffic.json; cp $P/this_pmc_hbm_counters.csv ${R}_pmc_hbm_counters.csv
cp $P/this_m15_traffic.json ${R}_m15_traffic.json; cp $P/this_m15_pmc_hbm_counters.csv ${R}_m15_pmc_hbm_counters.csv
cp $P/m15_bench.json ${R}_m15_bench.json; cp $P/m15_bench_under_rocprof.json ${R}_m15_bench_under_rocprof.json
cp $P/stats_m15/t_kernel_stats.csv ${R}_m15_kernel_stats_bench_steps2.csv
cp $P/cfg1_bench.json ${R}_cfg1_bench.json
cp $P/cfg3_bench_under_rocprof.json ${R}_cfg3_bench_under_rocprof.json; cp $P/stats_cfg3/t_kernel_stats.csv ${R}_cfg3_kernel_stats_bench_steps2.csv
cp $P/cfg4_bench_under_rocprof.json ${R}_cfg4_bench_under_rocprof.json; cp $P/stats_cfg4/t_kernel_stats.csv ${R}_cfg4_kernel_stats_bench_steps2.csv
# round 5: the counter tables of BASELINE configs[3] / [4], their bench lines with roofline.limiter, the wide-feature run
for c in cfg3 cfg4; do
  [ -f $P/this_${c}_traffic.json ] && cp $P/this_${c}_traffic.json ${R}_${c}_traffic.json && cp $P/this_${c}_pmc_hbm_counters.csv ${R}_${c}_pmc_hbm_counters.csv
  [ -f $P/${c}_bench.json ] && cp $P/${c}_bench.json ${R}_${c}_bench.json
done
[ -f $P/wide528_bench.json ] && cp $P/wide528_bench.json ${R}_wide528_bench.json
[ -f $P/this_wide528_traffic.json ] && cp $P/this_wide528_traffic.json ${R}_wide528_traffic.json && cp $P/this_wide528_pmc_hbm_counters.csv ${R}_wide528_pmc_hbm_counters.csv
[ -f $P/stats_wide528/t_kernel_stats.csv ] && cp $P/stats_wide528/t_kernel_stats.csv ${R}_wide528_kernel_stats_bench_steps2.csv && cp $P/wide528_bench_under_rocprof.json ${R}_wide528_bench_under_rocprof.json
