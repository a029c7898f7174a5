# Round-4 evidence under gpurun_out/r04p/ (copied into profiles/r04_* afterwards).  Run on the 1-GPU box, with the commit
# resolved HERE (the box has no .git):
#   gpurun --timeout 1100 -- "SRL_COMMIT=$(git rev-parse --short HEAD) bash tools/refresh_profiles_r04.sh"
# Counters are collected in their own passes (--kernel-trace + --pmc only), the program directly after `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p
rm -rf $O && mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err || exit 1
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu --no-dqn > $O/prof.log 2>&1 || exit 1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --no-cpu --no-dqn --steps 18 > $O/pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --no-cpu --no-dqn --steps 18 > $O/pmc_w.log 2>&1 || exit 1
python tools/pmc_summary.py $O/pmc_f $O/pmc_w srl_k_render 111656960 > $O/render_pmc.json || exit 1
echo render pmc done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_s1 -- python3 bench.py --no-cpu --no-dqn --steps 18 > $O/pmc_s1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_s2 -- python3 bench.py --no-cpu --no-dqn --steps 18 > $O/pmc_s2.log 2>&1 || exit 1
python tools/pmc_insts.py srl_k_step $O/pmc_s1 $O/pmc_s2 --json $O/settle_pmc.json > $O/settle_pmc.txt || exit 1
echo settle pmc done
rocprofv3 --kernel-trace --output-format csv -d $O/train -- python3 tools/profile_train.py run bf16x3 > $O/train.log 2>&1 || exit 1
(cd tools && python3 profile_train.py parse ../$O/train) > $O/dqn_update_profile.txt 2>&1
echo update profile done
for dt in fp32 bf16; do
  rocprofv3 --kernel-trace --output-format csv -d $O/qprof_$dt -- python3 tools/profile_qnet.py run 512 $dt > $O/qprof_$dt.log 2>&1 || exit 1
  python3 tools/profile_qnet.py parse $O/qprof_$dt > $O/qnet_rollout_$dt.txt 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_$dt -- python3 tools/profile_qnet.py run 512 $dt > $O/mfma_$dt.log 2>&1 || exit 1
  python3 tools/pmc_mfma.py $O/mfma_$dt --json $O/qnet_mfma_pmc_$dt.json > $O/qnet_mfma_pmc_$dt.txt 2>&1
done
echo rollout profiles done
python tools/bench_shapes.py > $O/env_shapes.txt 2>&1
python bench.py --config 2 --no-cpu --steps 17 --warmup 4 > $O/bench_config2.json 2> $O/bench_config2.err || exit 1
python bench.py --config 3 --no-cpu --steps 17 --warmup 4 > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
python bench.py --config 4 --no-cpu --steps 33 --warmup 4 > $O/bench_config4.json 2> $O/bench_config4.err || exit 1
python bench.py --steps 20 --warmup 5 --no-dqn --no-cpu > $O/bench_driver_window.json 2> $O/bench_driver_window.err || exit 1
python tools/stage_stats.py > $O/stage_stats.txt 2>&1
echo all done
