#!/usr/bin/env python3
"""Lock-step comparison of the one-handle loop and the group-wise loop (same seeds): where do they part?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('SRL_DIAG_QLIB'):
  from stackrl_amd import build as _b
  _b.QLIB = os.path.abspath(os.environ['SRL_DIAG_QLIB']); _b.qstale = lambda: False
import torch
from stackrl_amd import assets, env as envs, nets, qops
from stackrl_amd.dqn import DQN, PolynomialDecay
from stackrl_amd.training import Trainer
B, L = 512, 3
pool = assets.default_pool()
S = []
for groups in (None, 2):
  env = envs.make('Stack-v0', n_parallel=B, seed=5, pool=pool, episode_length=L, side_stream=True, **({} if groups is None else dict(groups=groups)))
  net = nets.DeepQSiamFCN(env.observation_spec, seed=2).cuda()
  agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=8, replay_memory_size=B * 8,
              discount_factor=.966667, collect_batch_size=B, exploration=0.5, prioritization=0.6,
              priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=9,
              policy_op=qops.FusedPolicy(chunk=256, fast=True), xcorr='bf16x3', prefetch=3, graphs=bool(int(os.environ.get('GRAPHS', '1'))))
  tr = Trainer(env, agent)
  tr.initialize(num_steps=2)
  seen = []
  orig = agent.observe
  agent._replay_memory_add = agent._replay_memory.add
  def add(state, reward, terminal, action, _s=seen, _a=agent):
    _s.append((state[0].clone(), state[1].clone(), reward.clone(), terminal.clone(), action.clone()))
    return _a._replay_memory_add(state, reward, terminal, action)
  agent._replay_memory.add = add
  S.append(dict(env=env, agent=agent, tr=tr, seen=seen, net=net, step=None))
for s in S:
  s['step'] = s['env'].reset(); s['agent'].acknowledge_reset()
for it in range(int(os.environ.get('ITERS', 2 * (L + 1) + 1))):
  out = []
  for s in S:
    s['step'] = s['tr'].collect_step(s['env'], s['step'])
    loss, _ = s['agent'].train()
    out.append(float(loss))
  a, b = S[0]['seen'][-1], S[1]['seen'][-1]
  names = ['obs_map', 'obs_obj', 'reward', 'terminal', 'action']
  diff = [n for n, x, y in zip(names, a, b) if not torch.equal(x, y)]
  wd = sum(int(not torch.equal(p, q)) for p, q in zip(S[0]['net'].parameters(), S[1]['net'].parameters()))
  print('it', it, 'differs:', diff, 'losses', out, 'params differing', wd)
  if 'action' in diff:
    bad = (a[4] != b[4]).nonzero()[:, 0].tolist()
    print('   action differs in envs', bad[:20], 'count', len(bad))
