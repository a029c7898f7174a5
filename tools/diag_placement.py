#!/usr/bin/env python3
"""Where do the settle workgroups' waves run?  Diagnostic build (SRL_STAMPS): every env's first two waves record HW_ID / XCC_ID;
prints, per CU, the envs resident together in the first round and the SIMD of each env's wave 0 / wave 1."""
import sys, os, subprocess, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stackrl_amd import build as B
so = os.path.join(ROOT, 'gpurun_out', 'libstackrl_stamps.so')
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(['/opt/rocm/bin/hipcc'] + B.FLAGS + ['-DSRL_STAMPS', os.path.join(B.CSRC, 'stackrl_hip.hip'), '-o', so])
B.LIB = so
import torch
from stackrl_amd import assets, env as envs, lib
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = envs.VecStackEnv(n_parallel=n, seed=11, pool=assets.default_pool(), block=True, episode_length=L)
g.reset()
for rep in range(3):
  g.step(g.sample())
  out = np.zeros((n, 2), np.int64)
  lib.load().srl_debug_hwid.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
  lib.load().srl_debug_hwid(g._h, out.ctypes.data_as(ctypes.c_void_p))
  hw, xcc = out & 0xffffffff, (out >> 32) & 0xf
  simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
  key = [(int(xcc[i, 0]), int(se[i, 0]), int(sh[i, 0]), int(cu[i, 0])) for i in range(n)]
  per = collections.defaultdict(list)
  for i in range(n): per[key[i]].append(i)
  pat = collections.Counter()
  share = 0
  for k, v in per.items():
    pat[tuple(sorted((int(simd[i, 0]), int(simd[i, 1])) for i in v))] += 1
    c = collections.Counter(int(simd[i, 0]) for i in v)
    share += sum(x - 1 for x in c.values() if x > 1)
  print('step %d: %d CUs hold envs; envs per CU: %s' % (rep, len(per), dict(collections.Counter(len(v) for v in per.values()))))
  print('  (wave-0 SIMD, wave-1 SIMD) patterns per CU, most common:', pat.most_common(6))
  print('  envs whose wave 0 shares its SIMD with another env\'s wave 0: %d of %d' % (share, n))
  ks = sorted(per)[:3]
  for k in ks: print('  CU', k, [(i, int(simd[i, 0]), int(simd[i, 1])) for i in per[k]])
