for rep in 1 2; do
  CHB_PACK_INCR=0 python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 --mix 0.5 --sigma 6e-3 > gpurun_out/r4_17_collapsed_rebuild_$rep.json 2>/dev/null
  python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 --mix 0.5 --sigma 6e-3 > gpurun_out/r4_17_collapsed_pack_$rep.json 2>/dev/null
  CHB_PACK_INCR=0 python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 --mix 0.3 --sigma 4.5e-3 > gpurun_out/r4_17_overlap_rebuild_$rep.json 2>/dev/null
  python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 --mix 0.3 --sigma 4.5e-3 > gpurun_out/r4_17_overlap_pack_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_17_*.json")):
    j=json.load(open(f)); k={x["kernel"]:round(x["ms_per_step"],2) for x in j["kernels"]}
    print(f.split("r4_17_")[1], round(j["ms_per_step"],2), k, j["fit_stats_last_call"])
PY
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "persistent_pack or segmented or labels_bit_exact" 2>&1 | tail -2
