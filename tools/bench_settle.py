#!/usr/bin/env python3
"""Settle kernel experiments: launch time vs solver iterations / env count (same seeds)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stackrl_amd import assets, env as envs
pool = assets.default_pool()
def run(B, L, iters, steps=None):
  g = envs.VecStackEnv(n_parallel=B, seed=11, pool=pool, block=False, episode_length=L, solver_iterations=iters)
  g.reset()()
  g.set_profiling(True)
  subs = []
  t0 = time.perf_counter()
  for k in range(steps or (L + 1) * 2):
    out = g.step(g.sample())
    if (k % (L + 1)) == L - 1:
      out(); subs.append(g.state()[2].sum(1))
  out()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  ms, n = g.kernel_times()
  s = np.concatenate(subs)
  print('B %d L %d iters %d: settle %.2f ms/launch, render %.1f us/launch; last-placement substeps mean %.1f max %d' % (
    B, L, iters, ms[0] / n[0], 1e3 * ms[1] / n[1], s.mean(), s.max()), flush=True)
  g.close()
for it in (10, 4, 1):
  run(1024, 8, it)
run(256, 8, 10)
run(4096, 8, 10)
run(512, 16, 10)
