#!/usr/bin/env python3
"""Golden vectors for the heuristic baseline policies from the reference's own `stackrl/baselines.py`.

Runs only in the build container.  baselines.py is numpy/scipy code, but its module header imports gin, gym and
`stackrl.agents` (TensorFlow), none of which is installed.  The functions exercised here (`height`, `difference`,
`corrcoef`, `correlate`, `goal_overlap`, `Baseline.call`) use none of those imports, so the module is loaded by file
path with inert placeholders for the three names (a no-op `gin.configurable`, an empty `gym`, a `PyGreedy` base
class that only forwards `__call__` to `call`).  Inputs are seeded uint8 observations in the env's format; the file
written (`baselines_golden.npz`) holds inputs and expected outputs only.
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True   # the reference tree is read-only: no __pycache__ beside its files
import types

import numpy as np

REF = '/root/reference/stackrl/baselines.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'baselines_golden.npz')


def load_reference():
  gin = types.ModuleType('gin')
  gin.configurable = lambda *a, **k: (lambda f: f) if not (len(a) == 1 and callable(a[0])) else a[0]
  gym = types.ModuleType('gym')
  stackrl = types.ModuleType('stackrl')
  agents = types.ModuleType('stackrl.agents')

  class PyGreedy(object):
    def __call__(self, inputs):
      return self.call(inputs)
  agents.PyGreedy = PyGreedy
  stackrl.agents = agents
  saved = {k: sys.modules.get(k) for k in ('gin', 'gym', 'stackrl', 'stackrl.agents')}
  sys.modules.update({'gin': gin, 'gym': gym, 'stackrl': stackrl, 'stackrl.agents': agents})
  try:
    spec = importlib.util.spec_from_file_location('ref_baselines', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
  finally:
    for k, v in saved.items():
      if v is None:
        sys.modules.pop(k, None)
      else:
        sys.modules[k] = v
  return mod


def make_obs(rng, H=128, h=32):
  """A plausible observation: blobs of height on the map, a goal rectangle, a rock-shaped object map."""
  m = np.zeros((H, H, 2), np.uint8)
  for _ in range(rng.randint(0, 6)):
    u, v = rng.randint(0, H - 30, 2); a, b = rng.randint(8, 30, 2)
    yy, xx = np.mgrid[0:a, 0:b]
    blob = (rng.randint(20, 90) * np.clip(1 - ((yy - a / 2) / (a / 2)) ** 2 - ((xx - b / 2) / (b / 2)) ** 2, 0, 1)).astype(np.uint8)
    m[u:u + a, v:v + b, 0] = np.maximum(m[u:u + a, v:v + b, 0], blob)
  gu, gv = rng.randint(8, 40, 2); gh, gw = rng.randint(32, 80), rng.randint(32, 80)
  m[gu:gu + gh, gv:gv + gw, 1] = 170
  o = np.zeros((h, h, 1), np.uint8)
  a, b = rng.randint(h // 3, h - 4, 2)
  yy, xx = np.mgrid[0:a, 0:b]
  rock = (60 - 35 * np.clip(1 - ((yy - a / 2) / (a / 2)) ** 2 - ((xx - b / 2) / (b / 2)) ** 2, 0, 1))
  rock = np.where(((yy - a / 2) / (a / 2)) ** 2 + ((xx - b / 2) / (b / 2)) ** 2 <= 1, rock, 0).astype(np.uint8)
  o[(h - a) // 2:(h - a) // 2 + a, (h - b) // 2:(h - b) // 2 + b, 0] = rock
  return m, o


def main():
  ref = load_reference()
  rng = np.random.RandomState(11)
  out = {}
  n_case = 5
  maps, objs = zip(*[make_obs(rng) for _ in range(n_case)])
  out['obs_map'] = np.stack(maps); out['obs_obj'] = np.stack(objs)
  for name, fn, kw in [('height', ref.height, {}), ('difference', ref.difference, {}),
                       ('difference_e1w0', ref.difference, dict(difference_exponent=1, weights_exponent=0)),
                       ('corrcoef', ref.corrcoef, {}), ('corrcoef_localized', ref.corrcoef, dict(localized=True)),
                       ('correlate', ref.correlate, {})]:
    out[name] = np.stack([np.asarray(fn((m.copy(), o.copy()), **kw), dtype=np.float64) for m, o in zip(maps, objs)])
  out['goal_overlap'] = np.stack([ref.goal_overlap((m, o)) for m, o in zip(maps, objs)])
  for method in ('height', 'difference', 'corrcoef', 'correlate'):
    for goal, minorder in ((True, 1), (True, 0), (False, 1)):
      pol = ref.Baseline(method=method, goal=goal, minorder=minorder)
      acts, vals = zip(*[pol((m.copy(), o.copy())) for m, o in zip(maps, objs)])
      tag = 'select_{}_g{}_m{}'.format(method, int(goal), minorder)
      out[tag + '_action'] = np.array(acts, dtype=np.int64)
      out[tag + '_values'] = np.stack([np.asarray(v, dtype=np.float64) for v in vals])
  np.savez_compressed(OUT, **out)
  print('wrote', OUT, os.path.getsize(OUT), 'bytes;', {k: v.shape for k, v in out.items() if not k.startswith('select')})


if __name__ == '__main__':
  main()
