#!/bin/bash
# Cross-compile the diagnostic build variants of libstackrl_hip.so into ab_libs/ (git-ignored; travels to the GPU box)
# for tools/ab_render.py --libs ab_libs and tools/pmc_ablate.sh.  Switches: csrc/render.hip, "Diagnostic builds only".
cd "$(dirname "$0")/.." || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-value"
mkdir -p ab_libs && rm -f ab_libs/lib*.so ab_libs/variants.txt
i=0
for v in "" "-DSRL_ABL_NOSTORE" "-DSRL_ABL_NOCAST" "-DSRL_ABL_NOSTAGE" "-DSRL_ABL_NOSTAGE -DSRL_ABL_NOGOAL" \
         "-DSRL_ABL_NOSTAGE -DSRL_ABL_NOGOAL -DSRL_ABL_NOTAIL -DSRL_ABL_NOOBJ -DSRL_ABL_NOSTORE" "-DSRL_ABL_EMPTY"; do
  /opt/rocm/bin/hipcc $F $v stackrl_amd/csrc/stackrl_hip.hip -o ab_libs/lib$i.so 2>/dev/null || exit 1
  echo "lib$i.so: ${v:-(product build)}" >> ab_libs/variants.txt
  i=$((i+1))
done
cat ab_libs/variants.txt
