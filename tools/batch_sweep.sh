#!/bin/bash
# developer run (through gpurun): sweep time against the speculative batch size at BASELINE configs[2] / [3] / [4]
# usage: bash tools/batch_sweep.sh <tag> [2|3|4 ...]
T=${1:-bsw}; shift
A="--steps 3 --warmup 1 --cpu-sample 0 --no-extra --no-e2e"
for cfg in "${@:-2 3 4}"; do
  case $cfg in
    2) C=""; BS="0 6144 10240 12288 16384";;
    3) C="--contigs 500000 --dim 140 --bins 128"; BS="0 24576 32768";;
    4) C="--contigs 1000000 --dim 146 --bins 200"; BS="0 24576 32768";;
  esac
  for b in $BS; do
    python bench.py $C --batch $b $A 2> gpurun_out/${T}_err.txt | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('configs[$cfg] batch $b: %.2f ms per sweep' % j['ms_per_step'], j.get('fit_stats_last_call'), {k['kernel']: round(k['ms_per_step'], 2) for k in j['kernels']})"
  done
done
