cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -- python3 bench.py --no-cpu > gpurun_out/prof3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_s1 -- python3 bench.py --no-cpu --steps 18 > gpurun_out/pmc_s1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_s2 -- python3 bench.py --no-cpu --steps 18 > gpurun_out/pmc_s2.log 2>&1 || exit 1
python tools/pmc_insts.py srl_k_step gpurun_out/pmc_s1 gpurun_out/pmc_s2 > gpurun_out/settle_pmc.txt
python tools/bench_shapes.py
