for rep in 1 2; do for k in 4096 6144 8192 12288; do
  python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 5 --batch $k > gpurun_out/r4_14_k${k}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_14_k*.json")):
    j=json.load(open(f)); k={x["kernel"]:round(x["ms_per_step"],2) for x in j["kernels"]}
    print(f.split("r4_14_")[1], round(j["ms_per_step"],3), "prefilter", k.get("prefilter"), "hull", k.get("hull_qp"), "bucket", k.get("bucket"), "upd", k.get("prefilter_update"), "slow", k.get("slow_path"), j["fit_stats_last_call"])
PY
