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

# per launch: is the slowest env's solve mostly ground phase?  (diag counters of the stamps build, cumulative over the L steps)
dg = np.zeros((n, 6), np.int64)
lib.load().srl_debug_diag.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.load().srl_debug_diag(g._h, dg.ctypes.data_as(ctypes.c_void_p))
order = np.argsort(-tot_sub)[:8]
print('the eight envs with the most sub-steps: sub-steps, of those without a pair point in wave 0, mean colours, mean solo sweeps per sub-step')
for i in order:
  print('  env %4d: %5d sub-steps, %5.1f %% without pair points, %.2f colours, %.1f sweeps, pair turns per sweep %.2f as scheduled / %.2f as the longest chain' % (i, dg[i, 0], 100.0 * dg[i, 1] / max(dg[i, 0], 1), dg[i, 2] / max(dg[i, 0], 1), dg[i, 3] / max(dg[i, 0], 1), dg[i, 4] / max(dg[i, 0], 1), dg[i, 5] / max(dg[i, 0], 1)))
print('all envs: %.1f %% of sub-steps without pair points, %.2f colours, %.1f sweeps, pair turns per sweep %.2f / %.2f' % (100.0 * dg[:, 1].sum() / dg[:, 0].sum(), dg[:, 2].sum() / dg[:, 0].sum(), dg[:, 3].sum() / dg[:, 0].sum(), dg[:, 4].sum() / dg[:, 0].sum(), dg[:, 5].sum() / dg[:, 0].sum()))
