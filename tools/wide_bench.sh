#!/bin/bash
# developer run (through gpurun): sweep times of the wide-row builds -- two, three, four slices; m = 5 and m = 15
# usage: bash tools/wide_bench.sh <tag>
T=${1:-wide}
A="--steps 2 --warmup 1 --cpu-sample 0 --no-extra --no-e2e"
for cfg in "200 5" "300 5" "528 5" "200 15" "300 15" "528 15"; do
  set -- $cfg
  python bench.py --dim $1 --neighbors $2 $A > gpurun_out/${T}_d$1_m$2.json 2> gpurun_out/${T}_err.txt || { tail -3 gpurun_out/${T}_err.txt; exit 1; }
  python - <<PY
import json
j = json.loads(open("gpurun_out/${T}_d$1_m$2.json").read().strip().splitlines()[-1])
print("D=$1 m=$2: %.2f ms per sweep" % j["ms_per_step"], {k["kernel"]: round(k["ms_per_step"], 2) for k in j["kernels"]})
PY
done
