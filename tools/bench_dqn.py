#!/usr/bin/env python3
"""DQN rollout + update timing (BASELINE configs[2] shape: B envs, L rocks, Q-net rollout + minibatch-32 update).
Reports iterations/s, env steps/s and the achieved FLOP/s of the Q-net forward against the dtype's gfx950 peak."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import assets, env as envs, nets, qops
from stackrl_amd.dqn import DQN, PolynomialDecay
from stackrl_amd.training import Trainer

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=1024)
ap.add_argument('--rocks', type=int, default=16)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--slots', type=int, default=16)
ap.add_argument('--side-stream', type=int, default=1)
ap.add_argument('--res', type=int, default=128, help='overhead map side (object map = res / 4): 128 = Stack-v0 default, 64 = BASELINE configs[4]')
ap.add_argument('--xcorr', default='bf16x3', help="update-path cross-correlation: bf16x3 | bf16 | library")
ap.add_argument('--graphs', type=int, default=0, help='replay the target evaluations of the update from hipGraphs')
ap.add_argument('--bf16', type=int, default=0, help='run the rollout forward under bf16 autocast (MFMA); fp32 is the reference dtype')
args = ap.parse_args()
B, L = args.envs, args.rocks
env = envs.make('Stack-v0', n_parallel=B, seed=11, pool=assets.default_pool(), episode_length=L, side_stream=bool(args.side_stream),
                **({} if args.res == 128 else dict(resolution_factor={64: 4, 256: 6}[args.res])))
net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=32, replay_memory_size=B * args.slots,
            discount_factor=.966667, collect_batch_size=B, exploration=PolynomialDecay(1.0, 400000, .1), prioritization=0.6,
            priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7,
            policy_op=qops.FusedPolicy(autocast=torch.bfloat16 if args.bf16 else None),
            xcorr=None if args.xcorr == 'library' else args.xcorr, graphs=bool(args.graphs))
tr = Trainer(env, agent)
tr.initialize(num_steps=4)
tr.run(2)
torch.cuda.synchronize()
# forward-only timing of the rollout policy
obs = env.reset()()[0]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): agent.policy(obs, exploration=True)
torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 3
macs = sum(nets.forward_macs(H=args.res, h=args.res // 4).values())
t0 = time.perf_counter(); tr.collect_time = tr.train_time = 0.0
tr.run(args.iters)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({'envs': B, 'rocks': L, 'iters_per_s': args.iters / dt, 'env_steps_per_s': args.iters * B / dt,
                  'rollout_forward_ms': tf * 1e3, 'rollout_forward_tflops': 2 * macs * B / tf / 1e12,
                  'rollout_dtype': 'bf16' if args.bf16 else 'f32',
                  'peak_tflops': 2500.0 if args.bf16 else 157.3, 'peak': 'dense bf16 MFMA' if args.bf16 else 'fp32 vector',
                  'frac_of_peak': 2 * macs * B / tf / ((2500.0 if args.bf16 else 157.3) * 1e12),
                  'update_xcorr': args.xcorr, 'graphs': bool(args.graphs), 'collect_s': tr.collect_time, 'train_s': tr.train_time}))
