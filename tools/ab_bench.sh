#!/bin/bash
# bench.py --no-cpu with each library of ab_libs/ in turn (twice): whole-step A/B of build variants on one box
cd $GRAFT_REPO_ROOT
cp stackrl_amd/libstackrl_hip.so /tmp/product.so
for rep in 1 2; do
  for so in ab_libs/lib*.so; do
    cp $so stackrl_amd/libstackrl_hip.so
    python bench.py --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'env_steps/s %.0f' % d['value'], 'settle %.3f ms' % d['settle']['avg_launch_ms'], 'render %.1f us' % d['roofline']['avg_launch_us'])"
  done
done
cp /tmp/product.so stackrl_amd/libstackrl_hip.so
