cd $GRAFT_REPO_ROOT
cp stackrl_amd/libstackrl_hip.so /tmp/product.so
for so in ab_libs/lib*.so; do
  cp $so stackrl_amd/libstackrl_hip.so
  echo "== $so"
  python tools/bench_shapes.py "$@" 2>/dev/null
done
cp /tmp/product.so stackrl_amd/libstackrl_hip.so
