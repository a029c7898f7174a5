# Round-2 evidence under gpurun_out/ (copied into profiles/r02_* afterwards).  Run on the 1-GPU box:
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles_r02.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err || exit 1
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof -- python3 bench.py --no-cpu --no-dqn > gpurun_out/r02_prof.log 2>&1 || exit 1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_pmc_f -- python3 bench.py --no-cpu --no-dqn --steps 18 > gpurun_out/r02_pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_pmc_w -- python3 bench.py --no-cpu --no-dqn --steps 18 > gpurun_out/r02_pmc_w.log 2>&1 || exit 1
python tools/pmc_summary.py gpurun_out/r02_pmc_f gpurun_out/r02_pmc_w srl_k_render 111656960 > gpurun_out/r02_render_pmc.json || exit 1
echo render pmc done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/r02_pmc_s1 -- python3 bench.py --no-cpu --no-dqn --steps 18 > gpurun_out/r02_pmc_s1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/r02_pmc_s2 -- python3 bench.py --no-cpu --no-dqn --steps 18 > gpurun_out/r02_pmc_s2.log 2>&1 || exit 1
python tools/pmc_insts.py srl_k_step gpurun_out/r02_pmc_s1 gpurun_out/r02_pmc_s2 > gpurun_out/r02_settle_pmc.txt
echo settle pmc done
python tools/bench_shapes.py > gpurun_out/r02_shapes.txt 2>&1
echo shapes done
rm -rf gpurun_out/r02_train
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02_train -- python3 tools/profile_train.py run bf16x3 > gpurun_out/r02_train.log 2>&1 || exit 1
(cd tools && python3 profile_train.py parse ../gpurun_out/r02_train) > gpurun_out/r02_dqn_update_profile.txt 2>&1
echo update profile done
for dt in fp32 bf16; do
  rm -rf gpurun_out/r02_qprof
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02_qprof -- python3 tools/profile_qnet.py run 512 $dt > gpurun_out/r02_qprof.log 2>&1 || exit 1
  python3 tools/profile_qnet.py parse gpurun_out/r02_qprof > gpurun_out/r02_qnet_rollout_$dt.txt 2>&1
done
rm -rf gpurun_out/r02_qprof
echo rollout profiles done
python bench.py --config 3 --no-cpu --steps 17 --warmup 4 > gpurun_out/r02_bench_config3.json 2> gpurun_out/r02_bench_config3.err || exit 1
python bench.py --config 4 --no-cpu --steps 33 --warmup 4 > gpurun_out/r02_bench_config4.json 2> gpurun_out/r02_bench_config4.err || exit 1
python bench.py --steps 20 --warmup 5 --no-dqn --no-cpu > gpurun_out/r02_bench_driver_window.json 2> gpurun_out/r02_bench_driver_window.err || exit 1
echo configs done
python tools/stamps.py 8 > gpurun_out/r02_stamps8.txt 2>&1
python tools/stamps.py 16 > gpurun_out/r02_stamps16.txt 2>&1
echo all done
