#!/usr/bin/env python3
"""Is the DQN update's gradient (hand-written convolutions, csrc/train_conv.hip) the same while env kernels run on another
stream?  One fixed minibatch, `DQN._forward_backward` repeated alone and under a stepping env; the flat gradient bucket and
the loss compared bit for bit with the first.  SRL_DIAG_QLIB selects the build of libstackrl_qnet.so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('SRL_DIAG_QLIB'):
  from stackrl_amd import build as _b
  _b.QLIB = os.path.abspath(os.environ['SRL_DIAG_QLIB']); _b.qstale = lambda: False
import torch
from stackrl_amd import assets, env as envs, nets, qops
from stackrl_amd.dqn import DQN, PolynomialDecay
from stackrl_amd.training import Trainer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
pool = assets.default_pool()
env = envs.make('Stack-v0', n_parallel=64, seed=3, pool=pool, episode_length=8)
net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=32, replay_memory_size=64 * 16,
            discount_factor=.966667, collect_batch_size=64, exploration=1.0, prioritization=0.6,
            priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7,
            policy_op=qops.FusedPolicy(fast=True), xcorr='bf16x3', graphs=False, prefetch=0)
tr = Trainer(env, agent)
tr.initialize(num_steps=12)
fixed = agent._next_minibatch()
fixed = (fixed[0].clone() if fixed[0] is not None else None, fixed[1].clone() if fixed[1] is not None else None,
         tuple(tuple(t.clone() for t in x) if isinstance(x, (tuple, list)) else x.clone() for x in fixed[2]))
agent._next_minibatch = lambda: fixed
load_env = envs.VecStackEnv(n_parallel=2048, seed=5, pool=pool, episode_length=8, side_stream=True)
load_env.reset()()
def run():
  loss = agent._forward_backward()[0]
  return agent._flat_grad.clone(), loss.clone()
g0, l0 = run(); torch.cuda.synchronize()
print('hand-written update path:', agent._hand is not None)
for load in ('none', 'env', 'none', 'env'):
  bad = 0
  for k in range(N):
    w = load_env.step(load_env.sample(), block=False) if load == 'env' else None
    g, l = run()
    torch.cuda.synchronize()
    if w is not None: w()
    if not (torch.equal(g.view(torch.int32), g0.view(torch.int32)) and torch.equal(l.view(torch.int32), l0.view(torch.int32))):
      bad += 1
      if bad <= 3:
        d = (g - g0).abs(); i = int(d.argmax())
        print('   repeat', k, ': gradient elements that differ', int((g.view(torch.int32) != g0.view(torch.int32)).sum()), 'largest |diff|', float(d.max()), 'at', i, 'value', float(g0[i]), 'loss', float(l), float(l0))
  print('load', load, ':', N, 'repeats; updates whose gradient or loss differ from the first:', bad, flush=True)
