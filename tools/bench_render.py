#!/usr/bin/env python3
"""Micro-benchmark of the render kernel (K2) on explicit poses: time vs number of rocks."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stackrl_amd import assets, env as envs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pool = assets.default_pool()
g = envs.VecStackEnv(n_parallel=B, seed=11, pool=pool, block=True, episode_length=8)
rng = np.random.RandomState(0)
for nbv in [0, 1, 2, 4, 8]:
  poses = np.zeros((B, 32, 7), np.float32); mesh = np.zeros((B, 32), np.int32)
  for b in range(nbv):
    q = rng.normal(size=(B, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses[:, b, 0] = rng.uniform(0.05, 0.45, B); poses[:, b, 1] = rng.uniform(0.05, 0.45, B); poses[:, b, 2] = rng.uniform(0.03, 0.12, B)
    poses[:, b, 3:] = q
    mesh[:, b] = rng.randint(len(pool), size=B)
  P = torch.from_numpy(poses).cuda(); M = torch.from_numpy(mesh).cuda(); N = torch.full((B,), nbv, dtype=torch.int32).cuda()
  out = torch.empty((B, 128, 128), dtype=torch.float32, device='cuda')
  for _ in range(3): g.render_heightmap(P, M, N, out)
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): g.render_heightmap(P, M, N, out)
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 20 * 1e3
  print('nb', nbv, 'us/launch %.1f' % us, 'coverage %.3f' % float((out > 0).float().mean()), flush=True)
