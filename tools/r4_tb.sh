export CHBIN_LIB=$PWD/ch-bin_amd/libchbin_hip_dev.so
for k in 2 3 4 6 100000; do
  CHB_SL_TILEK=$k python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 --neighbors 15 > gpurun_out/r4_29_m15_k${k}.json 2>/dev/null
done
CHB_SL_TILEK=3 python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 --neighbors 12 > gpurun_out/r4_29_m12_k3.json 2>/dev/null
python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 > gpurun_out/r4_29_m5_kdef.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_29_*.json")):
    j=json.load(open(f)); k={x["kernel"]:round(x["ms_per_step"],2) for x in j["kernels"]}
    print(f.split("r4_29_")[1], round(j["ms_per_step"],3), k)
PY
