#!/bin/bash
# Dynamic instruction counts of srl_k_render per ablation variant (ab_libs/lib*.so, cross-compiled with the switches
# listed in ab_libs/variants.txt): each variant replaces the in-tree library on the box's scratch copy of the repo.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp stackrl_amd/libstackrl_hip.so /tmp/product.so
for so in ab_libs/lib*.so; do
  i=$(basename $so .so)
  if grep "^$i.so" ab_libs/variants.txt | grep -q EMPTY; then continue; fi   # (its height maps are never written: the physics has nothing to stand on)
  cp $so stackrl_amd/libstackrl_hip.so
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py --no-cpu --steps 18 > gpurun_out/pmc_$i.log 2>&1 || exit 1
  echo "$i: $(grep "^$i.so" ab_libs/variants.txt | cut -d: -f2-)"; python tools/pmc_insts.py srl_k_render gpurun_out/pmc_$i | grep -E "per wave"
done
cp /tmp/product.so stackrl_amd/libstackrl_hip.so
