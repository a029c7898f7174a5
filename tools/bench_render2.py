#!/usr/bin/env python3
"""Render kernel experiments: where does the fixed per-launch cost of the first rock come from?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stackrl_amd import assets, env as envs
B = 1024
pool = assets.default_pool()
g = envs.VecStackEnv(n_parallel=B, seed=11, pool=pool, block=True, episode_length=8)
rng = np.random.RandomState(0)
def run(tag, nbv, xoff=0.0, same_mesh=False, z=None):
  poses = np.zeros((B, 32, 7), np.float32); mesh = np.zeros((B, 32), np.int32)
  for b in range(nbv):
    q = rng.normal(size=(B, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses[:, b, 0] = rng.uniform(0.05, 0.45, B) + xoff; poses[:, b, 1] = rng.uniform(0.05, 0.45, B); poses[:, b, 2] = rng.uniform(0.03, 0.12, B) if z is None else z
    poses[:, b, 3:] = q
    mesh[:, b] = 7 if same_mesh else rng.randint(len(pool), size=B)
  P = torch.from_numpy(poses).cuda(); M = torch.from_numpy(mesh).cuda(); N = torch.full((B,), nbv, dtype=torch.int32).cuda()
  out = torch.empty((B, 128, 128), dtype=torch.float32, device='cuda')
  for _ in range(3): g.render_heightmap(P, M, N, out)
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): g.render_heightmap(P, M, N, out)
  e1.record(); torch.cuda.synchronize()
  print(tag, 'nb', nbv, 'us/launch %.1f' % (e0.elapsed_time(e1) / 20 * 1e3), 'coverage %.3f' % float((out > 0).float().mean()), flush=True)
run('base', 0)
run('one rock', 1)
run('one rock off-map', 1, xoff=2.0)
run('one rock same mesh', 1, same_mesh=True)
run('one rock below ground', 1, z=-0.2)
run('8 rocks off-map', 8, xoff=2.0)
run('8 rocks same mesh', 8, same_mesh=True)
