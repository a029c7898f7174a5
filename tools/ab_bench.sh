#!/bin/bash
# bench.py (leg A, the driver's window) with each library of ab_libs/ in turn, REPS times round robin: whole-step A / B of build
# variants on ONE box (boxes differ by 1 - 2 % in the settle launch).  tools/build_variants.sh makes the libraries.
cd $GRAFT_REPO_ROOT
REPS=${REPS:-3}
cp stackrl_amd/libstackrl_hip.so /tmp/product.so
for rep in $(seq $REPS); do
  for so in ab_libs/lib*.so; do
    cp $so stackrl_amd/libstackrl_hip.so
    SRL_NO_FREE_RUN=${SRL_NO_FREE_RUN-1} python bench.py --no-cpu --no-dqn --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'env_steps/s %.0f' % d['value'], 'long %.0f' % d['value_long']['value'], 'settle %.4f ms' % d['settle']['avg_launch_ms'], 'render %.2f us' % d['roofline']['avg_launch_us'], 'ms/step %.4f' % d['ms_per_step'])"
  done
done
cp /tmp/product.so stackrl_amd/libstackrl_hip.so
