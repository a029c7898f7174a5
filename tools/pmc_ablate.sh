cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 0 1 2 3 4; do
  cp ab_libs/lib$i.so stackrl_amd/libstackrl_hip.so
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_ab$i -- python3 bench.py --no-cpu --steps 18 > gpurun_out/pmc_ab$i.log 2>&1 || exit 1
  echo variant $i; python tools/pmc_insts.py srl_k_render gpurun_out/pmc_ab$i | grep -E "INSTS_VALU|INSTS_SALU|INSTS_LDS"
done
