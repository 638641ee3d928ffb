#!/bin/bash
# Developer helper: build ch-bin_amd/libchbin_var_<name>.so with extra -D flags for ONE translation unit
# (default qp_kernels), reusing the product objects for the rest.  usage: build_variant.sh <name> "<flags>" [unit]
set -e
cd "$(dirname "$0")/../ch-bin_amd/csrc"
name=$1; flags=$2; unit=${3:-qp_kernels}
make -s -j6
mkdir -p _obj_var/$name
extra=""; [ "$unit" = topm_kernels ] && extra="-ffp-contract=off"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra $flags -c $unit.hip -o _obj_var/$name/$unit.o
objs=""
for u in topm_kernels prefilter_kernels qp_kernels aux_kernels kmer_kernels chb_api; do
  if [ $u = $unit ]; then objs="$objs _obj_var/$name/$u.o"; else objs="$objs _obj/$u.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libchbin_var_$name.so $objs
echo built ../libchbin_var_$name.so
