cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || exit 1
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -- python3 bench.py --no-cpu > gpurun_out/prof3.log 2>&1 || exit 1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --no-cpu --steps 18 > gpurun_out/pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --no-cpu --steps 18 > gpurun_out/pmc_w.log 2>&1 || exit 1
python tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w srl_k_render 111656960 > gpurun_out/render_pmc.json || exit 1
echo pmc done
python tools/stamps_render.py > gpurun_out/stamps.txt 2>&1 || exit 1
python tools/ab_render.py --libs ab_libs > gpurun_out/ablation.txt 2>&1 || exit 1
echo ablation done
bash tools/pmc_ablate.sh > gpurun_out/inst_counts.txt 2>&1 || exit 1
echo all done
# settle kernel counters (profiles/r01_settle_pmc.txt)
cd /tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_s1 -- python3 bench.py --no-cpu --steps 18 > gpurun_out/pmc_s1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_s2 -- python3 bench.py --no-cpu --steps 18 > gpurun_out/pmc_s2.log 2>&1 || exit 1
python tools/pmc_insts.py srl_k_step gpurun_out/pmc_s1 gpurun_out/pmc_s2 > gpurun_out/settle_pmc.txt
echo settle pmc done
