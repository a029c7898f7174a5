# Round-5 evidence under gpurun_out/r05p/ (copied into profiles/r05_* afterwards).  Run on the 1-GPU box, with the commit
# resolved HERE (the box has no .git):
#   gpurun --timeout 1100 -- "SRL_COMMIT=$(git rev-parse --short HEAD) bash tools/refresh_profiles_r05.sh [env|qnet|configs ...]"
# Counters are collected in their own passes (--kernel-trace + --pmc only), the program directly after `--`.
# What changed against round 4's script (VERDICT r04 item 2 / ADVICE): the profiling passes run WITHOUT the free-running leg
# (SRL_NO_FREE_RUN=1: every env-kernel launch of the pass is a 1,024-env launch) and the summaries select launches by grid
# (tools/pmc_summary.py --wgs, pmc_insts.py --wgs, kernel_stats.py: one row per launch size), so that profiles/ reproduces
# the line's avg_launch_us / traffic / valu_util.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p
mkdir -p $O
SECTIONS="${@:-env qnet configs}"
for S in $SECTIONS; do
case $S in
env)
  python bench.py > $O/bench_final.json 2> $O/bench_final.err || exit 1
  echo bench done
  python bench.py --steps 20 --warmup 5 --no-dqn > $O/bench_driver_window.json 2> $O/bench_driver_window.err || exit 1
  export SRL_NO_FREE_RUN=1
  rm -rf $O/prof $O/pmc_f $O/pmc_w $O/pmc_s1 $O/pmc_s2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu --no-dqn --steps 20 --warmup 5 > $O/prof.log 2>&1 || exit 1
  python tools/kernel_stats.py $O/prof --prefix srl_k > $O/bench_kernel_stats.csv || exit 1
  cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats_rocprofv3.csv
  echo stats done
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --no-cpu --no-dqn --steps 20 --warmup 5 > $O/pmc_f.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --no-cpu --no-dqn --steps 20 --warmup 5 > $O/pmc_w.log 2>&1 || exit 1
  python tools/pmc_summary.py $O/pmc_f $O/pmc_w srl_k_render 111656960 --wgs 1024 > $O/render_pmc.json || exit 1
  echo render pmc done
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_s1 -- python3 bench.py --no-cpu --no-dqn --steps 20 --warmup 5 > $O/pmc_s1.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_s2 -- python3 bench.py --no-cpu --no-dqn --steps 20 --warmup 5 > $O/pmc_s2.log 2>&1 || exit 1
  python tools/pmc_insts.py srl_k_step $O/pmc_s1 $O/pmc_s2 --wgs 1024 --json $O/settle_pmc.json > $O/settle_pmc.txt || exit 1
  echo settle pmc done
  # the settle kernel at a batch-bound shape (configs[2]'s 4,096 envs x 16 rocks: the two-wave variant srl_k_step_t128, ordered
  # launch): the same counters
  rm -rf $O/pmc_b1 $O/pmc_b2
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_b1 -- python3 bench.py --no-cpu --no-dqn --envs 4096 --rocks 16 --steps 17 --warmup 4 > $O/pmc_b1.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_b2 -- python3 bench.py --no-cpu --no-dqn --envs 4096 --rocks 16 --steps 17 --warmup 4 > $O/pmc_b2.log 2>&1 || exit 1
  python tools/pmc_insts.py srl_k_step_t128 $O/pmc_b1 $O/pmc_b2 --wgs 4096 --json $O/settle_pmc_4096x16.json > $O/settle_pmc_4096x16.txt || exit 1
  unset SRL_NO_FREE_RUN
  python tools/argmax_precision.py 4608 1 > $O/rollout_argmax.json 2> $O/rollout_argmax.err
  ;;
qnet)
  rm -rf $O/train $O/qprof_* $O/mfma_*
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
  ;;
configs)
  python tools/bench_shapes.py > $O/env_shapes.txt 2>&1
  python bench.py --config 2 --no-cpu --steps 17 --warmup 4 > $O/bench_config2.json 2> $O/bench_config2.err || exit 1
  python bench.py --config 3 --no-cpu --steps 17 --warmup 4 > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
  python bench.py --config 4 --no-cpu --steps 33 --warmup 4 > $O/bench_config4.json 2> $O/bench_config4.err || exit 1
  ;;
esac
done
echo all done
