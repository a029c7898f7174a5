# DQN configs[2] under experiment hooks of bench.py (one line per setting).  On the GPU box: bash tools/experiments/cfg2_knobs.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04/exp
run() { name=$1; shift; env "$@" python bench.py --config 2 --no-cpu --steps 17 --warmup 4 > gpurun_out/r04/exp/$name.json 2> gpurun_out/r04/exp/$name.err; python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/r04/exp/$name.json').read().strip().splitlines()[-1]); q=d['dqn']['bf16x3']
  print('%-28s %8.0f /s  %6.2f ms/iter  fwd in loop %6.2f  alone %6.2f' % ('$name', q['env_steps_per_s'], q['ms_per_iter'], q['rollout_forward_ms'], q.get('rollout_forward_alone_ms', 0)), flush=True)
except Exception as e:
  print('$name', 'failed', e, open('gpurun_out/r04/exp/$name.err').read()[-600:])
PY
}
run base A=1
F=ffffffff
run mask_low224 SRL_FWD_CU_MASK=$F,$F,$F,$F,$F,$F,$F,00000000
run mask_low192 SRL_FWD_CU_MASK=$F,$F,$F,$F,$F,$F,00000000,00000000
run mask_low128 SRL_FWD_CU_MASK=$F,$F,$F,$F,00000000,00000000,00000000,00000000
run mask_spread224 SRL_FWD_CU_MASK=fefefefe,fefefefe,fefefefe,fefefefe,fefefefe,fefefefe,fefefefe,fefefefe
run mask_spread192 SRL_FWD_CU_MASK=eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee
run mask_all SRL_FWD_CU_MASK=$F,$F,$F,$F,$F,$F,$F,$F
