#!/bin/bash
# Collect the round's profile set on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into profiles/).
# usage (through gpurun): bash tools/collect_profiles.sh r05prof [quick | a | b | w]
#   (the whole set no longer fits one 20-minute gpurun call: `a` = configs[2] and m = 15, `b` = configs[1] / [3] / [4] with
#    their counter tables and the wide-feature run; both write into the same gpurun_out/<tag>/)
#   kernel stats (rocprofv3 --kernel-trace --stats) + the bench line of the same run, HBM-side counters (separate --pmc
#   passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes), the default bench line, and the other configs.
set -o pipefail
tag=${1:-r05prof}; quick=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
PB="--steps 1 --warmup 0 --cpu-sample 0 --no-e2e --no-extra"
if [ "$quick" != "b" ] && [ "$quick" != "w" ]; then
echo "[1] kernel stats, config 2"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -o t -- $B --steps 3 --warmup 1 --cpu-sample 0 --no-e2e --no-extra > $O/bench_under_rocprof.json 2> $O/err_stats_cfg2.txt || exit 1
echo "[2] pmc fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o t -- $B --steps 1 --warmup 0 --cpu-sample 0 --no-e2e --no-extra > $O/bench_pmc_fetch.json 2> $O/err_pmc_fetch.txt || exit 1
echo "[3] pmc write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o t -- $B --steps 1 --warmup 0 --cpu-sample 0 --no-e2e --no-extra > $O/bench_pmc_write.json 2> $O/err_pmc_write.txt || exit 1
# on-chip side of the dominant kernels (roofline.limiter of the bench line): L2 -> L1 read requests, L2 hit rate, busy shares
echo "[3b] pmc tcp / tcc / sq"
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/pmc_tcp -o t -- $B $PB > $O/bench_pmc_tcp.json 2> $O/err_pmc_tcp.txt || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/pmc_tcc -o t -- $B $PB > $O/bench_pmc_tcc.json 2> $O/err_pmc_tcc.txt || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/pmc_sq -o t -- $B $PB > $O/bench_pmc_sq.json 2> $O/err_pmc_sq.txt || exit 1
if [ "$quick" != "quick" ]; then
  for p in tcp:"TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" tcc:"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" sq:"SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" fetch:FETCH_SIZE write:WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d $O/m15/pmc_${p%%:*} -o t -- $B --neighbors 15 $PB > $O/m15_bench_pmc_${p%%:*}.json 2> $O/m15_err_pmc_${p%%:*}.txt || exit 1
  done
  python3 $R/tools/traffic_from_pmc.py $O/m15 $O/this_m15 > $O/traffic_m15_log.txt 2>&1 || exit 1
fi
# (the default line quotes the traffic table of THIS build: bench.py only quotes a table whose source stamp matches)
python3 $R/tools/traffic_from_pmc.py $O $O/this --exclude m15 > $O/traffic_log.txt 2>&1 || exit 1
echo "[4] default bench"; $B --traffic-file $O/this_traffic.json > $O/bench_default.json 2> $O/err_default.txt || exit 1
[ "$quick" = "quick" ] && exit 0
echo "[5] m = 15"; $B --neighbors 15 --steps 3 --warmup 1 --no-extra --traffic-file $O/this_m15_traffic.json > $O/m15_bench.json 2> $O/err_m15.txt || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_m15 -o t -- $B --neighbors 15 --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-extra > $O/m15_bench_under_rocprof.json 2> $O/err_stats_m15.txt || exit 1
[ "$quick" = "a" ] && exit 0
fi
if [ "$quick" != "w" ]; then
echo "[6] configs[1]"; $B --contigs 10000 --bins 32 --steps 5 --warmup 2 --cpu-sample 200 --no-extra > $O/cfg1_bench.json 2> $O/err_cfg1.txt || exit 1
echo "[7] configs[3]"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3 -o t -- $B --contigs 500000 --dim 140 --bins 128 --steps 2 --warmup 1 --cpu-sample 0 --no-extra > $O/cfg3_bench_under_rocprof.json 2> $O/err_cfg3.txt || exit 1
echo "[8] configs[4]"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o t -- $B --contigs 1000000 --dim 146 --bins 200 --steps 2 --warmup 1 --cpu-sample 0 --no-extra > $O/cfg4_bench_under_rocprof.json 2> $O/err_cfg4.txt || exit 1
# counters of the kernels that dominate the multi-GPU configurations (their own tables: bench.py quotes
# r05_cfg3_traffic.json / r05_cfg4_traffic.json for --contigs 500000 ... / 1000000 ...)
for c in cfg3:"--contigs 500000 --dim 140 --bins 128" cfg4:"--contigs 1000000 --dim 146 --bins 200"; do
  name=${c%%:*}; cargs=${c#*:}
  echo "[9] pmc $name"
  for p in tcp:"TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" tcc:"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" sq:"SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" fetch:FETCH_SIZE write:WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d $O/$name/pmc_${p%%:*} -o t -- $B $cargs $PB > $O/${name}_bench_pmc_${p%%:*}.json 2> $O/${name}_err_pmc_${p%%:*}.txt || exit 1
    find $O/$name -name "*kernel_trace.csv" -delete
  done
  python3 $R/tools/traffic_from_pmc.py $O/$name $O/this_$name > $O/traffic_${name}_log.txt 2>&1 || exit 1
  $B $cargs --steps 2 --warmup 1 --cpu-sample 0 --no-extra --traffic-file $O/this_${name}_traffic.json > $O/${name}_bench.json 2> $O/err_${name}_bench.txt || exit 1
done
fi   # (w: the wide-feature part only)
# VERDICT r4 item 8, "first, the numbers": a feature width beyond the shortlist stage's 157 columns (k = 5: 512 k-mer columns)
echo "[10] wide features"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wide528 -o t -- $B --dim 528 --steps 2 --warmup 1 --cpu-sample 0 --no-extra --no-e2e > $O/wide528_bench_under_rocprof.json 2> $O/err_stats_wide528.txt || exit 1
for p in tcp:"TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" tcc:"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" sq:"SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" fetch:FETCH_SIZE write:WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc ${p#*:} --output-format csv -d $O/wide528/pmc_${p%%:*} -o t -- $B --dim 528 $PB > $O/wide528_bench_pmc_${p%%:*}.json 2> $O/wide528_err_pmc_${p%%:*}.txt || exit 1
  find $O/wide528 -name "*kernel_trace.csv" -delete
done
python3 $R/tools/traffic_from_pmc.py $O/wide528 $O/this_wide528 > $O/traffic_wide528_log.txt 2>&1 || exit 1
$B --dim 528 --steps 3 --warmup 1 --cpu-sample 0 --no-extra --no-e2e --traffic-file $O/this_wide528_traffic.json > $O/wide528_bench.json 2> $O/err_wide528.txt || exit 1
find $O -name "*kernel_trace.csv" -size +8M -delete
echo done
