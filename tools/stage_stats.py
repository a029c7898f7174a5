#!/usr/bin/env python3
"""Statistics of the staged render records (srl_k_stage) over an episode at the headline shape: items per rock against its
bounding box, and the sizes of the list ranges an item sweeps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stackrl_amd import assets, env as envs
B, L = 256, 8
g = envs.VecStackEnv(n_parallel=B, seed=11, pool=assets.default_pool(), block=True, episode_length=L)
g.reset()
for t in range(L):
  g.step(g.sample())
  r = g.stage_records().view(np.int32)
  nb = t + 1
  hd0, hd1 = r[:, :nb, 0], r[:, :nb, 1]
  i0, i1, j0, j1 = hd0[..., 0] & 0xffff, hd0[..., 0] >> 16, hd0[..., 1] & 0xffff, hd0[..., 1] >> 16
  nup, nsil, items, nir = hd0[..., 2], hd0[..., 3], hd1[..., 0], hd1[..., 1]
  bbox_items = ((i1 - i0 + 4) // 4) * ((j1 - j0 + 2) // 2)
  span = r[:, :nb, 2:6].reshape(B, nb, 16); rng = r[:, :nb, 6:10].reshape(B, nb, 16).view(np.uint32)
  start = span >> 8
  cnt = np.diff(np.concatenate([start, items[..., None]], -1), axis=-1)          # items per row
  pn = ((rng >> 8) & 0xff).astype(int) - (rng & 0xff).astype(int); sn = (rng >> 24).astype(int) - ((rng >> 16) & 0xff).astype(int)
  w = np.maximum(cnt, 0)
  print('rocks %d: items/rock %.0f (bbox %.0f)  nup %.1f nsil %.1f  planes swept per item %.1f  sides per item %.1f  total items/env %.0f' % (
    nb, items.mean(), bbox_items.mean(), nup.mean(), nsil.mean(), (pn * w).sum() / w.sum(), (sn * w).sum() / w.sum(), items.sum(1).mean()), flush=True)
