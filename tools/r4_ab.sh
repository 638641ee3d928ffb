# developer A/B helper of round 4 (run through gpurun): variants in ch-bin_amd/libchbin_var_<name>.so against the product library
# usage: bash tools/r4_ab.sh <tag> "<variants>" "<configs: cfg2 m15 cfg3 cfg4>" [reps]
tag=$1; vars=$2; cfgs=$3; reps=${4:-2}
O=gpurun_out
B="python bench.py --no-extra --cpu-sample 0 --no-e2e"
for rep in $(seq 1 $reps); do
for c in $cfgs; do
  case $c in
    cfg2) A="--steps 5" ;;
    m15) A="--steps 3 --neighbors 15" ;;
    cfg3) A="--steps 3 --contigs 500000 --dim 140 --bins 128" ;;
    cfg4) A="--steps 2 --contigs 1000000 --dim 146 --bins 200" ;;
  esac
  for v in $vars; do
    if [ $v = prod ]; then unset CHBIN_LIB; else export CHBIN_LIB=$PWD/ch-bin_amd/libchbin_var_$v.so; fi
    $B $A > $O/${tag}_${c}_${v}_$rep.json 2>$O/${tag}_${c}_${v}_$rep.err || echo fail $c $v
  done
done; done
unset CHBIN_LIB
python - $tag <<'PY'
import json,glob,sys
for f in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}_*.json")):
    try:
        j=json.load(open(f)); k={x["kernel"]:round(x["ms_per_step"],2) for x in j["kernels"]}
        print(f.split(sys.argv[1]+"_")[1], round(j["ms_per_step"],2), "prefilter", k.get("prefilter"), "hull", k.get("hull_qp"), "bucket", k.get("bucket"), "upd", k.get("prefilter_update"), "slow", k.get("slow_path"))
    except Exception as e: print(f, "ERR", e)
PY
