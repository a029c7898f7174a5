#!/usr/bin/env python3
"""Two identical handles advanced one raw sub-step at a time (srl_step_simulation): A while the convolution kernels run on
the current stream, B alone.  Prints the first sub-steps after which their velocities / poses differ, and by how much."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('SRL_DIAG_LIB'):
  from stackrl_amd import build as _b
  _b.LIB = os.path.abspath(os.environ['SRL_DIAG_LIB']); _b.stale = lambda: False
import numpy as np, torch
from stackrl_amd import assets, env as envs, nets, qops
B, L, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
PER = int(sys.argv[4]) if len(sys.argv) > 4 else 1     # raw sub-steps per launch
pool = assets.default_pool()
A = envs.VecStackEnv(n_parallel=B, seed=5, pool=pool, episode_length=L, side_stream=True)
Bv = envs.VecStackEnv(n_parallel=B, seed=5, pool=pool, episode_length=L, side_stream=True)
net = nets.DeepQSiamFCN(A.observation_spec, seed=2).cuda()
ff = qops.FastFeatures(net, dtype=torch.float32)
C = torch.randn(256, 1, 97, 97, device='cuda')
with torch.no_grad(): ff.pos(C)
A.reset()(); Bv.reset()()
for t in range(L - 1):                      # fill the scene, no load
  a = A.sample(); b = Bv.sample()
  assert torch.equal(a, b)
  A.step(a)(); Bv.step(b)()
assert all(np.array_equal(x, y) for x, y in zip(A.state(), Bv.state())), 'the two handles differ before any load'
# lift the newest rock a little so that things move again
pa = A.state()[0].copy(); nb = A.state()[1]
for e in range(B):
  pa[e, nb[e] - 1, 2] += 0.02
A.set_body_state(poses=pa); Bv.set_body_state(poses=pa)
found = 0
for k in range(N):
  with torch.no_grad():                       # the load is queued first; A's launch (its own stream) lands in the middle of it
    for _ in range(6): ff.pos(C)
  envs._check(A._lib.srl_step_simulation(A._h, PER, A._stream()))
  with torch.no_grad():
    for _ in range(3 if PER == 1 else 12): ff.pos(C)
  torch.cuda.synchronize()
  envs._check(Bv._lib.srl_step_simulation(Bv._h, PER, Bv._stream()))
  torch.cuda.synchronize()
  va, vb = A.velocities(), Bv.velocities()
  sa, sb = A.state()[0], Bv.state()[0]
  dv = (va.view(np.uint32) != vb.view(np.uint32)); dp = (sa.view(np.uint32) != sb.view(np.uint32))
  if dv.any() or dp.any():
    found += 1
    es = np.nonzero(dv.reshape(B, -1).any(1) | dp.reshape(B, -1).any(1))[0]
    print('sub-step', k, ': differ in envs', es[:8].tolist())
    e = es[0]
    print('   sweeps A', A.sweeps()[es[:8]].tolist(), 'B', Bv.sweeps()[es[:8]].tolist(), 'contacts A', [x[es[:8]].tolist() for x in A.contacts()], 'B', [x[es[:8]].tolist() for x in Bv.contacts()])
    bod = np.nonzero(dv[e].any(1) | dp[e].any(1))[0]
    for b_ in bod[:3]:
      print('   env', e, 'body', b_, 'nb', nb[e])
      print('      vel A', va[e, b_].tolist()); print('      vel B', vb[e, b_].tolist())
      print('      pose A', sa[e, b_].tolist()); print('      pose B', sb[e, b_].tolist())
    # resynchronise A to B so that the next difference is a fresh one
    A.set_body_state(poses=sb, velocities=vb)
    if found >= 1: break
  if PER > 1 and k % 5 == 4:                 # keep things moving: lift the newest rock again in both
    pa = Bv.state()[0].copy()
    for e in range(B):
      pa[e, nb[e] - 1, 2] += 0.02
    A.set_body_state(poses=pa); Bv.set_body_state(poses=pa)
print('done', k + 1, 'sub-steps,', found, 'with a difference')
