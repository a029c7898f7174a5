#!/usr/bin/env python3
"""Are the Q-net's kernels' results the same while env kernels run on another stream?  The rollout forward (features,
cross-correlation, position head) and one DQN update on fixed inputs, repeated alone and under a stepping env; every result
compared bit for bit with the first.  SRL_DIAG_QLIB selects the build of libstackrl_qnet.so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('SRL_DIAG_QLIB'):
  from stackrl_amd import build as _b
  _b.QLIB = os.path.abspath(os.environ['SRL_DIAG_QLIB']); _b.qstale = lambda: False
import numpy as np, torch
from stackrl_amd import assets, env as envs, nets, qops, qtrain
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pool = assets.default_pool()
e = envs.VecStackEnv(n_parallel=2048, seed=5, pool=pool, episode_length=8, side_stream=True)
e.reset()()
net = nets.DeepQSiamFCN(e.observation_spec, seed=2).cuda()
ff = qops.FastFeatures(net, dtype=torch.float32)
g = torch.Generator(device='cuda').manual_seed(1)
xm = torch.randint(0, 256, (256, 128, 128, 2), device='cuda', dtype=torch.uint8, generator=g)
xo = torch.randint(0, 256, (256, 32, 32, 1), device='cuda', dtype=torch.uint8, generator=g)
hand = qtrain.HandNet(net)
xm2, xo2 = xm[:64].contiguous(), xo[:64].contiguous()
def fwd():
  with torch.no_grad():
    X, W = ff((xm, xo)); C = qops.xcorr_forward(X, W); A = ff.pos(C)
  return A
def upd():
  q = hand.forward((xm2, xo2), save=True)
  gq = torch.ones_like(q) * 1e-3
  hand.backward(gq)
  return torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None]) if any(p.grad is not None for p in net.parameters()) else q
ref_f = fwd().clone()
try:
  ref_u = upd().clone(); has_u = True
except Exception as x:
  print('update leg skipped:', repr(x)[:120]); has_u = False
torch.cuda.synchronize()
for load in ('none', 'env', 'none'):
  bf = bu = 0
  for k in range(N):
    w = None
    if load == 'env':
      w = e.step(e.sample(), block=False)
    a = fwd()
    u = upd() if has_u else None
    torch.cuda.synchronize()
    if w is not None: w()
    bf += int(not torch.equal(a.view(torch.int32), ref_f.view(torch.int32)))
    if has_u: bu += int(not torch.equal(u.view(torch.int32), ref_u.view(torch.int32)))
  print('load', load, ':', N, 'repeats; forward results that differ from the first:', bf, '; update results that differ:', bu, flush=True)
