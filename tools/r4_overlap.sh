export CHBIN_LIB=$PWD/ch-bin_amd/libchbin_hip_dev.so
for rep in 1 2; do for o in 0 1 2; do
  CHB_DEV_OVERLAP=$o python bench.py --no-extra --cpu-sample 0 --no-e2e --steps 5 > gpurun_out/r4_19_overlap${o}_$rep.json 2>gpurun_out/r4_19_overlap${o}_$rep.err || tail -3 gpurun_out/r4_19_overlap${o}_$rep.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_19_overlap*.json")):
    j=json.load(open(f)); k={x["kernel"]:round(x["ms_per_step"],2) for x in j["kernels"]}
    print(f.split("r4_19_")[1], round(j["ms_per_step"],3), k)
PY
