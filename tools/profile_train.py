#!/usr/bin/env python3
"""Per-kernel breakdown of the steady-state DQN minibatch update (`DQN.train`, minibatch 32).

`rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/profile_train.py run`, then
`python3 tools/profile_train.py parse <dir>` (kernels between the two arange markers)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ITERS = 3


def run(B=256, L=8):
  import torch
  from stackrl_amd import assets, env as envs, nets, qops
  from stackrl_amd.dqn import DQN, PolynomialDecay
  from stackrl_amd.training import Trainer
  env = envs.make('Stack-v0', n_parallel=B, seed=11, pool=assets.default_pool(), episode_length=L)
  net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
  agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=32, replay_memory_size=B * 16,
              discount_factor=.966667, collect_batch_size=B, exploration=PolynomialDecay(1.0, 400000, .1), prioritization=0.6,
              priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7,
              policy_op=qops.FusedPolicy(autocast=torch.bfloat16), xcorr=(sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != 'library' else None))
  tr = Trainer(env, agent)
  tr.initialize(num_steps=4)
  for _ in range(3): agent.train()
  torch.cuda.synchronize()
  torch.arange(12345, device='cuda'); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(ITERS): agent.train()
  torch.cuda.synchronize()
  print('train step wall: %.2f ms' % ((time.perf_counter() - t0) / ITERS * 1e3))
  torch.arange(23456, device='cuda'); torch.cuda.synchronize()


if __name__ == '__main__':
  if sys.argv[1] == 'run':
    run()
  else:
    import profile_qnet
    profile_qnet.parse(sys.argv[2])
