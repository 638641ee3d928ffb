#!/bin/bash
# developer A/B (through gpurun): the product library against a variant library on the default, overlapping-bins and
# collapsed-bins workloads.  usage: bash tools/ab_lib.sh <variant .so relative to the repo>
V=$PWD/$1
A="--steps 3 --warmup 1 --no-extra --no-e2e --cpu-sample 0"
for w in "" "--mix 0.3 --sigma 0.0045" "--mix 0.5 --sigma 0.006"; do
  for lib in default $V; do
    if [ $lib = default ]; then unset CHBIN_LIB; else export CHBIN_LIB=$lib; fi
    python bench.py $w $A 2>/dev/null | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$w]', '$lib'.split('/')[-1], '%.2f ms' % j['ms_per_step'], j.get('fit_stats_last_call'), {k['kernel']: round(k['ms_per_step'], 2) for k in j['kernels']})"
  done
done
