#!/usr/bin/env python3
"""Diagnostic for round 2's anomaly (a): with the env step on its side stream the LIBRARY bias-gradient
reduction of the 256-channel bottom layers returned garbage in element 128 (5 of 6 runs of configs[4] within ten updates).
Round 2 replaced the library passes by hand-written ones without naming a cause.  This script attributes it: the update
runs eagerly with the library bias / ReLU path (fused epilogues off) while the side stream carries, by argv[1],

  env        the env step (settle + render kernels of libstackrl_hip.so), as in `Training.run`
  torchload  NOT the env but a chain of large library matrix products (no kernel of this repo on the side stream);
             the env step itself runs to completion before the update starts
  serial     nothing: env step and update take turns

and reports the first corrupt gradient element (parameter, index, bit pattern).  `env` corrupt + `torchload` clean puts
the cause in this repo's env kernels or their buffers; `torchload` corrupt puts it in the library reduction under any
concurrent kernel.  SRL_DIAG_LIB selects another libstackrl_hip.so.  Always exits 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

if os.environ.get('SRL_DIAG_LIB'):
  from stackrl_amd import build as _b
  _b.LIB = os.path.abspath(os.environ['SRL_DIAG_LIB'])

from stackrl_amd import assets, env as envs, nets, qops
from stackrl_amd.dqn import DQN, PolynomialDecay
from stackrl_amd.training import Trainer

mode = sys.argv[1] if len(sys.argv) > 1 else 'env'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 14
fused = len(sys.argv) > 3 and sys.argv[3] == 'fused'
B, L = 2048, 32
env = envs.make('Stack-v0', n_parallel=B, seed=11, pool=assets.default_pool(), episode_length=L, side_stream=(mode == 'env'),
                resolution_factor=4)
net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=32, replay_memory_size=B * 16,
            discount_factor=.966667, collect_batch_size=B, exploration=PolynomialDecay(1.0, 400000, .1), prioritization=0.6,
            priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7,
            policy_op=qops.FusedPolicy(autocast=torch.bfloat16, fast=True), xcorr='bf16x3', graphs=False)
if not fused:
  net.set_fused_epilogues(False)              # the library bias add / ReLU / bias-gradient reduction of round 2's failing runs
print('mode', mode, 'fused epilogues', net.fused_epilogues, flush=True)
tr = Trainer(env, agent)
tr.initialize(num_steps=4)
side = torch.cuda.Stream()
wa = torch.randn(6144, 6144, device='cuda')
wb = torch.randn(6144, 6144, device='cuda')
step = env.reset()
agent.acknowledge_reset()
names = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
nbad = 0
for it in range(iters):
  if callable(step):
    step = step()
  action = agent.collect(*step)
  step = env.step(action)
  if mode != 'env':
    step = step() if callable(step) else step
    torch.cuda.synchronize()
  if mode == 'torchload':
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      x = wa
      for _ in range(10):
        x = (x @ wb) * 1e-2
  loss, mtd = agent.train()
  if mode == 'torchload':
    torch.cuda.current_stream().wait_stream(side)
  g = agent._flat_grad
  bad = ((g.abs() > 1e4) | ~torch.isfinite(g)).nonzero()[:, 0]
  print('update %2d loss %.5g |g|max %.4g bad %d' % (it, float(loss), float(g.abs().max()), bad.numel()), flush=True)
  if bad.numel():
    nbad += 1
    for n, p in names:
      off = (p.grad.data_ptr() - g.data_ptr()) // 4
      inside = bad[(bad >= off) & (bad < off + p.numel())]
      if inside.numel():
        idx = (inside - off).tolist()[:8]
        vals = g[inside[:8]].view(torch.int32).tolist()
        print('   corrupt: %s shape %s elements %s bits %s' % (n, tuple(p.shape), idx, ['%08x' % (v & 0xffffffff) for v in vals]), flush=True)
    g.zero_()
    break
step() if callable(step) else None
print('diag_reduce mode=%s fused=%s: %s' % (mode, net.fused_epilogues, 'CORRUPT GRADIENT' if nbad else 'clean over %d updates' % iters), flush=True)
env.close()
