#!/bin/bash
# developer A/B (through gpurun): the 16-lane solver's full-set start against the grown corral (a library built with
# -DCHB_QP16_BULK=0 in tools/_bin/), the solver's tests, and the fp64 matrix-core rate microbenchmark
# usage: bash tools/ab_m15.sh <tag>
T=${1:-r5_m15}
O=gpurun_out
[ -x tools/_bin/mfma_f64_rate ] && tools/_bin/mfma_f64_rate > $O/${T}_mfma.txt 2>&1 && cat $O/${T}_mfma.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "16_lane or vs_oracle_and_enumerator or kkt or affine or more_than_16" > $O/${T}_tests.txt 2>&1
tail -3 $O/${T}_tests.txt
A="--neighbors 15 --steps 3 --warmup 1 --no-extra --no-e2e --cpu-sample 0"
for i in 1 2; do
  python bench.py $A > $O/${T}_bulk_$i.json 2> $O/${T}_err.txt
  CHBIN_LIB=$PWD/tools/_bin/libchbin_hip_nobulk.so python bench.py $A > $O/${T}_grow_$i.json 2>> $O/${T}_err.txt
done
python - <<PY
import json
for n in ("bulk_1", "grow_1", "bulk_2", "grow_2"):
    try:
        j = json.loads(open("$O/${T}_%s.json" % n).read().strip().splitlines()[-1])
        ks = j.get("kernels")
        print(n, round(j["ms_per_step"], 2), {k["kernel"]: round(k["ms_per_step"], 2) for k in ks})
    except Exception as e:
        print(n, "failed", e)
PY
