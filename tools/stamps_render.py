#!/usr/bin/env python3
"""Diagnostic build: phase shares of the render kernel (thread 0 of every workgroup), by rocks present."""
import sys, os, subprocess, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stackrl_amd import build as B
so = os.path.join(ROOT, 'gpurun_out', 'libstackrl_stamps.so')
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(['/opt/rocm/bin/hipcc'] + B.FLAGS + sys.argv[1:] + ['-DSRL_STAMPS', os.path.join(B.CSRC, 'stackrl_hip.hip'), '-o', so])
B.LIB = so
import torch
from stackrl_amd import assets, env as envs, lib
pool = assets.default_pool()
n, L = 1024, 8
g = envs.VecStackEnv(n_parallel=n, seed=11, pool=pool, block=True, episode_length=L)
fn = lib.load().srl_debug_rstamps; fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = np.zeros((n, 8), np.int64)
g.reset(); fn(g._h, out.ctypes.data_as(ctypes.c_void_p), 1)
names = ['prologue (hdr, poses, mesh hdr)', 'bounds+staging', 'early-stores', 'pixel pass', '(loop end)', 'epilogue', 'tail (object obs, sum tree, reward)']
print('mean ns per workgroup by phase; thread-0 wall clock (100 MHz)')
for k in range(L):
  g.step(g.sample()); fn(g._h, out.ctypes.data_as(ctypes.c_void_p), 1)
  per = out.mean(0) * 10.0
  print('nb %d: ' % (k + 1) + ', '.join('%s %.0f' % (nm.split(' ')[0], v) for nm, v in zip(names, per)) + '  | total %.1f us' % (per[:7].sum() / 1e3))
  if k + 1 == L:   # first wave of workgroups (resident at launch) against the ones that follow them
    for nm_, sl in (('workgroups 0-511  ', slice(0, n // 2)), ('workgroups 512-1023', slice(n // 2, n))):
      pr = out[sl].mean(0) * 10.0
      print('  ' + nm_ + ': ' + ', '.join('%.0f' % v for v in pr[:7]) + '  | total %.1f us' % (pr[:7].sum() / 1e3))
