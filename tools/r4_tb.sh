export CHBIN_LIB=$PWD/ch-bin_amd/libchbin_hip_dev.so
for m in 8 12; do for tb in 16 100000; do
  CHB_SL_TILEBEST=$tb python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 --neighbors $m > gpurun_out/r4_25_m${m}_tb${tb}.json 2>/dev/null
done; done
for tb in 16 64 100000; do
  CHB_SL_TILEBEST=$tb python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 > gpurun_out/r4_25_m5_tb${tb}.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_25_m*.json")):
    j=json.load(open(f)); k={x["kernel"]:round(x["ms_per_step"],2) for x in j["kernels"]}
    print(f.split("r4_25_")[1], round(j["ms_per_step"],3), k)
PY
