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
L = int(sys.argv[4]) if len(sys.argv) > 4 else 16
res = int(sys.argv[5]) if len(sys.argv) > 5 else 128
env = envs.make('Stack-v0', n_parallel=B, seed=11, pool=assets.default_pool(), episode_length=L, side_stream=True,
                **(dict(resolution_factor=4) if res == 64 else {}))
net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
if os.environ.get('NOBIASGRAD'):
  for nm, p_ in net.named_parameters():
    if 'bottom' in nm and nm.endswith('bias'):
      p_.requires_grad_(False)
agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=32, replay_memory_size=B * 16,
            discount_factor=.966667, collect_batch_size=B, exploration=PolynomialDecay(1.0, 400000, .1), prioritization=0.6,
            priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7,
            policy_op=qops.FusedPolicy(autocast=torch.bfloat16 if dtype == 'bf16' else None, fast=True), xcorr='bf16x3', graphs=graphs)
tr = Trainer(env, agent)
tr.initialize(num_steps=4)
step = env.reset(); agent.acknowledge_reset()
for it in range(int(os.environ.get('ITERS', 14))):
  if callable(step): step = step()
  action = agent.collect(*step)
  bad = int(((action < 0) | (action >= env.n_actions)).sum())
  print(it, 'actions bad', bad, 'min', int(action.min()), 'max', int(action.max()), flush=True)
  if bad:
    idx = ((action < 0) | (action >= env.n_actions)).nonzero()[:8, 0]
    print('   bad envs', idx.tolist(), 'actions', action[idx].tolist(), flush=True)
    break
  if os.environ.get('SERIAL'):
    torch.cuda.synchronize()
    watch = {'flat_grad': agent._flat_grad, 'flat': agent._optimizer.flat, 'm': agent._optimizer.m, 'v': agent._optimizer.v}
    snap = {k: t.clone() for k, t in watch.items()}
  step = env.step(action)
  if os.environ.get('SERIAL'):
    step = step() if callable(step) else step
    torch.cuda.synchronize()
    for k, t in watch.items():
      d = (t.view(torch.int32) != snap[k].view(torch.int32)).nonzero()[:, 0]
      if d.numel():
        print('   ENV STEP CHANGED %s: %d words, first %s -> %s' % (k, d.numel(), d[:12].tolist(),
              ['%08x' % (x & 0xffffffff) for x in t.view(torch.int32)[d[:12]].tolist()]), flush=True)
  loss, mtd = agent.train()
  gmax = float(agent._flat_grad.abs().max())
  if not (gmax < 1e6):
    fg = agent._flat_grad
    bad_idx = ((fg.abs() > 1e6) | ~torch.isfinite(fg)).nonzero()[:, 0]
    print('   BAD GRAD: %d elements, first %s values %s' % (bad_idx.numel(), bad_idx[:8].tolist(), fg[bad_idx[:8]].tolist()), flush=True)
    o = 0
    lay = agent._optimizer.layout(list(net.parameters())) if hasattr(agent._optimizer, 'layout') else None
    for (name, p_), off in zip(net.named_parameters(), lay[0] if lay else []):
      inside = ((bad_idx >= off) & (bad_idx < off + p_.numel())).sum().item()
      if inside:
        print('     in %s shape %s offset %d: %d bad' % (name, tuple(p_.shape), off, inside), flush=True)
  fl = agent._optimizer.flat
  print('   loss %.5g mtd %.5g  params finite %s  grad finite %s  |g|max %.3g  m finite %s v min %.3g lr_t %s' % (
    float(loss), float(mtd), bool(torch.isfinite(fl).all()), bool(torch.isfinite(agent._flat_grad).all()),
    float(agent._flat_grad.abs().max()), bool(torch.isfinite(agent._optimizer.m).all()), float(agent._optimizer.v.min()),
    agent._optimizer.state.tolist()), flush=True)
step() if callable(step) else None
print('ok')
