#!/usr/bin/env python3
"""Serial against early-gradient update (hipGraph replay), iteration by iteration: where do the weights part?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import assets, env as envs, nets, qops
from stackrl_amd.dqn import DQN, PolynomialDecay
from stackrl_amd.training import Trainer
B, L = 64, 3
pool = assets.default_pool()
graphs = bool(int(os.environ.get('GRAPHS', '1')))
hist = []
for early in (False, True):
  env = envs.make('Stack-v0', n_parallel=B, seed=5, pool=pool, episode_length=L, side_stream=True)
  net = nets.DeepQSiamFCN(env.observation_spec, seed=2).cuda()
  agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=8, replay_memory_size=B * 8,
              discount_factor=.966667, collect_batch_size=B, exploration=0.5, prioritization=0.6, target_update_period=4,
              priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=9,
              policy_op=qops.FusedPolicy(fast=True), xcorr='bf16x3', prefetch=2, graphs=graphs, early_gradient=early)
  tr = Trainer(env, agent)
  tr.initialize(num_steps=2)
  h = []
  step = env.reset(); agent.acknowledge_reset()
  for it in range(int(os.environ.get('ITERS', '14'))):
    agent.train_begin()
    step = tr.collect_step(env, step)
    loss, _ = agent.train()
    torch.cuda.synchronize()
    h.append((float(loss), float(agent._flat_grad.double().abs().sum()), float(sum(p.double().abs().sum() for p in net.parameters())),
              agent._last_sample_indexes.tolist(), float(agent._replay_memory._logits[torch.isfinite(agent._replay_memory._logits)].double().sum())))
  tr._drain(env, step)
  hist.append(h); env.close()
for it, (a, b) in enumerate(zip(*hist)):
  flags = ['loss', 'grad', 'params', 'indexes', 'priorities']
  d = [f for f, x, y in zip(flags, a, b) if x != y]
  print('it', it, 'differs:', d, (a[:3], b[:3]) if (d or it >= 7) else '')
