export CHBIN_LIB=$PWD/ch-bin_amd/libchbin_hip_dev.so
for rep in 1 2; do for r in 0.6 0.8 1.0 1.5 2.5; do
  CHB_EARLY_RATIO=$r python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 5 > gpurun_out/r4_23_ratio${r}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_23_ratio*.json")):
    j=json.load(open(f)); k={x["kernel"]:round(x["ms_per_step"],2) for x in j["kernels"]}
    print(f.split("r4_23_")[1], round(j["ms_per_step"],3), j["fit_stats_last_call"]["batches"], k)
PY
