#!/usr/bin/env python3
"""The graph-replayed serial update against the eager one, iteration by iteration: which parameters' gradients differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import assets, env as envs, nets, qops
from stackrl_amd.dqn import DQN, PolynomialDecay
from stackrl_amd.training import Trainer
B, L = 64, 3
pool = assets.default_pool()
hist = []
names = None
for graphs in (False, True):
  env = envs.make('Stack-v0', n_parallel=B, seed=5, pool=pool, episode_length=L, side_stream=True)
  net = nets.DeepQSiamFCN(env.observation_spec, seed=2).cuda()
  agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=8, replay_memory_size=B * int(os.environ.get('SLOTS', '8')),
              discount_factor=.966667, collect_batch_size=B, exploration=0.5, prioritization=0.6, target_update_period=int(os.environ.get('TUP', '4')),
              priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=9,
              policy_op=qops.FusedPolicy(fast=True), xcorr='bf16x3', prefetch=2, graphs=graphs)
  names = [n for n, _ in net.named_parameters()]
  tr = Trainer(env, agent)
  tr.initialize(num_steps=2)
  h = []
  step = env.reset(); agent.acknowledge_reset()
  for it in range(12):
    step = tr.collect_step(env, step)
    loss, _ = agent.train()
    torch.cuda.synchronize()
    h.append((float(loss.detach()), [p.grad.detach().clone() for p in net.parameters()], agent._last_sample_indexes.clone(), [p.detach().clone() for p in net.parameters()], agent._flat_grad.clone()))
  tr._drain(env, step)
  hist.append(h); env.close()
for it, (a, b) in enumerate(zip(*hist)):
  bad = [(n, float((x - y).abs().max()), float(x.abs().max())) for n, x, y in zip(names, a[1], b[1]) if not torch.equal(x, y)]
  badp = [n for n, x, y in zip(names, a[3], b[3]) if not torch.equal(x, y)]
  fd = (a[4] != b[4]).nonzero()[:, 0]
  print('it', it, 'params that differ', len(badp), badp[:4], 'flat-grad elements that differ', int(fd.numel()), fd[:6].tolist(), 'of', a[4].numel(), 'loss', a[0] == b[0], 'indexes', torch.equal(a[2], b[2]), 'params whose gradient differs:', len(bad), bad[:6])
