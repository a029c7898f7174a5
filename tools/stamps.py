#!/usr/bin/env python3
"""Diagnostic build with in-kernel stamps: where does a settle sub-step spend its time?
(Separate .so; shares are read, never the run time of this build.)"""
import sys, os, subprocess, ctypes
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
pool = assets.default_pool()
n, L = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = envs.VecStackEnv(n_parallel=n, seed=11, pool=pool, block=True, episode_length=L)
g.reset()
tot_sub = 0
for k in range(L):
  g.step(g.sample()); tot_sub += g.state()[2].sum(1)
out = np.zeros((n, 12), np.int64)
lib.load().srl_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.load().srl_debug_stamps(g._h, out.ctypes.data_as(ctypes.c_void_p))
names = ['frame', 'verts', 'bounds+ground', 'broadphase+colour', 'narrowphase', 'point setup', 'solver', 'integrate']
per = out[:, :8].sum(0) / tot_sub.sum() * 10.0   # 100 MHz ticks -> ns
np2 = out[:, 8:].sum(0)
slow = np.argmax(tot_sub)
print('mean ns per sub-step by phase (all envs):')
for nm, v in zip(names, per): print('  %-20s %8.0f ns  %5.1f%%' % (nm, v, 100 * v / per.sum()))
ps = out[slow, :8] / tot_sub[slow] * 10.0
print('slowest env (%d sub-steps), ns per sub-step by phase:' % tot_sub[slow])
for nm, v in zip(names, ps): print('  %-20s %8.0f ns  %5.1f%%' % (nm, v, 100 * v / ps.sum()))
print('  total %.1f us per sub-step' % (ps.sum() / 1e3))
print('  total %.1f us; slowest env: %d sub-steps, %.2f ms' % (per.sum() / 1e3, tot_sub[slow], out[slow, :8].sum() * 1e-5))
print('  slot-0 narrowphase per call: refresh %.0f ns, gjk %.0f ns, insert %.0f ns (%d calls, %.2f per sub-step)' % (10 * np2[0] / np2[3], 10 * np2[1] / np2[3], 10 * np2[2] / np2[3], np2[3], np2[3] / tot_sub.sum()))
