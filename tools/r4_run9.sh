python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "persistent_pack or golden_flow or labels_bit_exact or odd_shapes or small_batch" 2>&1 | tail -2
for rep in 1 2 3; do
  CHB_PACK_INCR=0 python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 5 > gpurun_out/r4_11_cfg2_rebuild_$rep.json 2>/dev/null
  python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 5 > gpurun_out/r4_11_cfg2_pack_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_11_*.json")):
    try:
        j=json.load(open(f)); k={x["kernel"]:round(x["ms_per_step"],2) for x in j["kernels"]}
        print(f.split("r4_11_")[1], round(j["ms_per_step"],3), "prefilter", k.get("prefilter"), "hull", k.get("hull_qp"), "bucket", k.get("bucket"), "upd", k.get("prefilter_update"), "slow", k.get("slow_path"))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_11_stats -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-e2e --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r4_11_rocprof.json 2>/dev/null
