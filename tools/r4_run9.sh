python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r4_09_tests.txt 2>&1; echo tests rc=$?; tail -3 gpurun_out/r4_09_tests.txt
for rep in 1 2 3; do
  CHB_PACK_INCR=0 python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 5 > gpurun_out/r4_09_cfg2_rebuild_$rep.json 2>/dev/null
  python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 5 > gpurun_out/r4_09_cfg2_pack_$rep.json 2>/dev/null
done
python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 --neighbors 15 > gpurun_out/r4_09_m15_pack_1.json 2>/dev/null
python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 3 --contigs 500000 --dim 140 --bins 128 > gpurun_out/r4_09_cfg3_pack_1.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_09_*.json")):
    try:
        j=json.load(open(f)); k={x["kernel"]:round(x["ms_per_step"],2) for x in j["kernels"]}
        print(f.split("r4_09_")[1], round(j["ms_per_step"],3), "prefilter", k.get("prefilter"), "hull", k.get("hull_qp"), "bucket", k.get("bucket"), "upd", k.get("prefilter_update"), "slow", k.get("slow_path"))
    except Exception as e: print(f, "ERR", e)
PY
