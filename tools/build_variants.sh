#!/bin/bash
# Build variants of the env library THE PRODUCT'S WAY (vectoriser + isa_fix pass, stackrl_amd/build.py) into ab_libs/ for a
# same-box A / B with tools/ab_bench.sh:  tools/build_variants.sh name1 "flags1" name2 "flags2" ...
cd "$(dirname "$0")/.." || exit 1
mkdir -p ab_libs && rm -f ab_libs/lib*.so ab_libs/variants.txt
cp stackrl_amd/libstackrl_hip.so /tmp/product_keep.so
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  SRL_EXTRA_FLAGS="$flags" python -c "
from stackrl_amd import build as b
b.build(force=True) if False else b._build_env_fixed('/opt/rocm/bin/hipcc', False)" || exit 1
  cp stackrl_amd/libstackrl_hip.so ab_libs/lib_$name.so
  echo "lib_$name.so: ${flags:-(product build)}" >> ab_libs/variants.txt
done
cp /tmp/product_keep.so stackrl_amd/libstackrl_hip.so
cat ab_libs/variants.txt
