#!/usr/bin/env python3
"""Experiment: rollout forward of env group g under the straggler tail of the other groups' settle kernels.

B envs as G handles of B / G envs on G side streams.  Per iteration and group: the policy forward of the group waits (on the
device) for the group's own step, then the group's next step is launched — either at once (`eager`) or after the forwards
of every group (`deferred`).  G = 1 is today's loop.  Prints ms per iteration over one episode (L + 1 calls)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--rocks', type=int, default=16)
ap.add_argument('--groups', type=int, default=4)
ap.add_argument('--mode', default='deferred')
ap.add_argument('--prio', type=int, default=0, help='1: env streams at high priority')
ap.add_argument('--bf16', type=int, default=0)
ap.add_argument('--episodes', type=int, default=1)
ap.add_argument('--chunk', type=int, default=2048)
args = ap.parse_args()
os.environ.setdefault('SRL_STEP_VARIANT', 'two_wave' if args.envs >= 3072 else 'four_wave')
import torch
from stackrl_amd import assets, env as envs, nets, qops

B, L, G = args.envs, args.rocks, args.groups
n = B // G
pool = assets.default_pool()
main = torch.cuda.current_stream()
es = [envs.VecStackEnv(n_parallel=n, seed=11, pool=pool, episode_length=L, side_stream=True, env_index_offset=k * n) for k in range(G)]
if args.prio:
  for e in es:
    e._side = torch.cuda.Stream(priority=-1)
e0 = es[0]
net = nets.DeepQSiamFCN(e0.observation_spec, seed=1).cuda()
pol = qops.FusedPolicy(chunk=args.chunk, autocast=torch.bfloat16 if args.bf16 else None, fast=True)
gen = torch.Generator(device='cuda'); gen.manual_seed(3)


def outputs():
  return (torch.empty((B, 128, 128, 2), dtype=torch.uint8, device='cuda'), torch.empty((B, 32, 32, 1), dtype=torch.uint8, device='cuda'),
          torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'))


bufs = [outputs(), outputs()]
sl = lambda ts, k: tuple(t[k * n:(k + 1) * n] for t in ts)
done = [None] * G
for k, e in enumerate(es):
  e.reset(block=False, out=sl(bufs[0][:2], k))
  done[k] = torch.cuda.Event(); done[k].record(e._side)
cur = 0
throttle = []


def iteration():
  global cur
  nxt = 1 - cur
  acts = []
  for k, e in enumerate(es):
    main.wait_event(done[k])
    om, oo = sl(bufs[cur][:2], k)
    a = pol(net, (om, oo), 1.0, gen)
    acts.append(a)
    if args.mode == 'eager':
      e.step(a, block=False, out=sl(bufs[nxt], k))
      done[k] = torch.cuda.Event(); done[k].record(e._side)
  if args.mode != 'eager':
    for k, e in enumerate(es):
      e.step(acts[k], block=False, out=sl(bufs[nxt], k))
      done[k] = torch.cuda.Event(); done[k].record(e._side)
  cur = nxt
  ev = torch.cuda.Event(); ev.record(main)
  throttle.append(ev)
  if len(throttle) > 2:
    throttle.pop(0).synchronize()


for _ in range(L + 1):
  iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = args.episodes * (L + 1)
for _ in range(N):
  iteration()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
status = 'ok'
for e in es:
  try:
    envs._check(e._lib.srl_sync_status(e._h, e._stream()))
  except RuntimeError as x:
    status = str(x)[:60]
print(json.dumps({'envs': B, 'rocks': L, 'groups': G, 'mode': args.mode, 'prio': args.prio, 'bf16': args.bf16, 'chunk': args.chunk,
                  'status': status, 'ms_per_iter': 1e3 * dt / N, 'placements_per_s': B * L * args.episodes / dt}))
