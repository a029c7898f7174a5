#!/usr/bin/env python3
"""Diagnostic: the DQN loop of bench.py's leg B with finiteness checks after every piece."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import assets, env as envs, nets, qops
from stackrl_amd.dqn import DQN, PolynomialDecay
from stackrl_amd.training import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f32'
graphs = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
env = envs.make('Stack-v0', n_parallel=B, seed=11, pool=assets.default_pool(), episode_length=16, side_stream=True)
net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=32, replay_memory_size=B * 16,
            discount_factor=.966667, collect_batch_size=B, exploration=PolynomialDecay(1.0, 400000, .1), prioritization=0.6,
            priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7,
            policy_op=qops.FusedPolicy(autocast=torch.bfloat16 if dtype == 'bf16' else None), xcorr='bf16x3', graphs=graphs)
tr = Trainer(env, agent)
tr.initialize(num_steps=4)
step = env.reset(); agent.acknowledge_reset()
for it in range(14):
  if callable(step): step = step()
  action = agent.collect(*step)
  bad = int(((action < 0) | (action >= env.n_actions)).sum())
  print(it, 'actions bad', bad, 'min', int(action.min()), 'max', int(action.max()), flush=True)
  step = env.step(action)
  loss, mtd = agent.train()
  fl = agent._optimizer.flat
  print('   loss %.5g mtd %.5g  params finite %s  grad finite %s  |g|max %.3g  m finite %s v min %.3g lr_t %s' % (
    float(loss), float(mtd), bool(torch.isfinite(fl).all()), bool(torch.isfinite(agent._flat_grad).all()),
    float(agent._flat_grad.abs().max()), bool(torch.isfinite(agent._optimizer.m).all()), float(agent._optimizer.v.min()),
    agent._optimizer.state.tolist()), flush=True)
step() if callable(step) else None
print('ok')
