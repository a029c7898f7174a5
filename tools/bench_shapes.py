#!/usr/bin/env python3
"""Env-only throughput (random policy on device) at the BASELINE shapes: envs x rocks x height-map size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stackrl_amd import assets, env as envs
pool = assets.default_pool()
shapes = [(1024, 8, 5), (1024, 16, 5), (4096, 16, 5), (2048, 32, 4)]
if len(sys.argv) > 1:
  shapes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]]
# LAUNCH_ORDER=0 | 1: settle workgroups in index order / highest release first (srl_set_launch_order); unset: by batch size
lo = os.environ.get('LAUNCH_ORDER')
okw = {} if lo is None else {'launch_order': bool(int(lo))}
for B, L, rf in shapes:
  g = envs.VecStackEnv(n_parallel=B, seed=11, pool=pool, block=False, episode_length=L, resolution_factor=rf, **okw)
  g.reset()()
  for _ in range(L + 1): out = g.step(g.sample())
  out(); torch.cuda.synchronize()
  g.kernel_times(); g.set_profiling(True)
  t0 = time.perf_counter()
  reps = 2 if B * L > 40000 else 4
  for _ in range(reps * (L + 1)): out = g.step(g.sample())
  out(); torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  ms, n = g.kernel_times()
  print('%5d envs x %2d rocks, %3d^2 map: %8.0f env steps/s  (settle %.2f ms, render %.1f us per launch)' % (
    B, L, 4 * 2 ** rf, B * L * reps / dt, ms[0] / n[0], 1e3 * ms[1] / n[1]), flush=True)
  g.close()
