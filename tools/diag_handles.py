#!/usr/bin/env python3
"""Diagnostic for round 2's anomaly (b) ("What's weak"): several env handles stepping on their own streams
returned observations that differed from the one-handle env when the scenario ran at the end of tests/test_learner_gpu.py
(gpurun_out/t_learner.log), and agreed when it ran alone.  This script replays that order in ONE process — first the
configs[4] loop with the graph-replayed update under the side-stream env step (the test that ran before it), then the
comparison — and, unlike the deleted test, says WHICH tensor of WHICH env differed at WHICH call and what the two envs'
internal states (mesh ids, goal rectangles, sub-step counts, poses) look like at that point.

  SRL_DIAG_LIB=<path to another libstackrl_hip.so>   run against that build (e.g. round 2's, built from git)
  HEAVY=0                                            skip the configs[4] prelude
Always exits 0; the findings are on stdout."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

if os.environ.get('SRL_DIAG_LIB'):
  from stackrl_amd import build as _b
  _b.LIB = os.path.abspath(os.environ['SRL_DIAG_LIB'])
  print('library under test:', _b.LIB, flush=True)

from stackrl_amd import assets, env as envs, nets, qops
from stackrl_amd.dqn import DQN, PolynomialDecay
from stackrl_amd.training import Trainer


def heavy_prelude():
  B, L = 2048, 32
  env = envs.make('Stack-v0', n_parallel=B, seed=11, pool=assets.default_pool(), episode_length=L, side_stream=True,
                  resolution_factor=4)
  net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
  agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=32, replay_memory_size=B * 16,
              discount_factor=.966667, collect_batch_size=B, exploration=PolynomialDecay(1.0, 400000, .1), prioritization=0.6,
              priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7,
              policy_op=qops.FusedPolicy(autocast=torch.bfloat16, fast=True), xcorr='bf16x3', graphs=True)
  tr = Trainer(env, agent)
  tr.initialize(num_steps=4)
  step = env.reset()
  agent.acknowledge_reset()
  for it in range(8):
    if callable(step):
      step = step()
    action = agent.collect(*step)
    step = env.step(action)
    agent.train()
  step() if callable(step) else None
  g = agent._flat_grad
  print('prelude: 8 iterations of configs[4], gradient finite', bool(torch.isfinite(g).all()), 'max', float(g.abs().max()), flush=True)
  env.close()


def describe(tag, a, shards, G):
  """Internal state of the one-handle env `a` against the shards'."""
  pa, nba, suba, sta = a.state()
  Ha, Oa, ga = a.maps()
  ps = [s.state() for s in shards]
  ms = [s.maps() for s in shards]
  pb = np.concatenate([p[0] for p in ps]); nbb = np.concatenate([p[1] for p in ps]); subb = np.concatenate([p[2] for p in ps])
  Hb = np.concatenate([m[0] for m in ms]); gb = np.concatenate([m[2] for m in ms])
  print('   [%s] n_bodies differ in envs %s' % (tag, np.nonzero(nba != nbb)[0][:16].tolist()))
  print('   [%s] sub-step counts differ in envs %s' % (tag, np.nonzero((suba != subb).any(1))[0][:16].tolist()))
  gd = np.nonzero((ga != gb).any(1))[0]
  print('   [%s] goal rectangles differ in envs %s  e.g. %s vs %s' % (tag, gd[:16].tolist(), ga[gd[:2]].tolist(), gb[gd[:2]].tolist()))
  md = np.nonzero((pa[..., 7] != pb[..., 7]).any(1))[0]
  print('   [%s] mesh ids differ in envs %s' % (tag, md[:16].tolist()))
  pd = np.abs(pa[..., :7] - pb[..., :7]).reshape(len(pa), -1).max(1)
  print('   [%s] poses differ in envs %s  max %g' % (tag, np.nonzero(pd > 0)[0][:16].tolist(), float(pd.max())))
  hd = np.abs(Ha - Hb).reshape(len(Ha), -1).max(1)
  print('   [%s] height maps differ in envs %s  max %g' % (tag, np.nonzero(hd > 0)[0][:16].tolist(), float(hd.max())), flush=True)


def compare(pool, B, L, K, seed, rounds):
  G = B // K
  a = envs.VecStackEnv(n_parallel=B, seed=seed, pool=pool, episode_length=L)
  shards = [envs.VecStackEnv(n_parallel=G, seed=seed, pool=pool, episode_length=L, env_index_offset=k * G, side_stream=True)
            for k in range(K)]
  assert a.seed(seed) == sum((s.seed(seed) for s in shards), [])
  filler = torch.randn(2048, 2048, device='cuda')

  def shard_outputs():
    om = torch.empty((B, a._H, a._H, 2), dtype=torch.uint8, device='cuda')
    oo = torch.empty((B,) + tuple(a.observation_spec[1].shape), dtype=torch.uint8, device='cuda')
    r = torch.empty(B, dtype=torch.float32, device='cuda')
    d = torch.empty(B, dtype=torch.uint8, device='cuda')
    return om, oo, r, d

  om, oo, _, _ = shard_outputs()
  waits = [s.reset(block=False, out=(om[k * G:(k + 1) * G], oo[k * G:(k + 1) * G])) for k, s in enumerate(shards)]
  sa = a.reset()()
  for w in waits:
    w()
  sb = ((om, oo), torch.zeros(B, device='cuda'), torch.zeros(B, dtype=torch.bool, device='cuda'))
  bad = 0
  for t in range(rounds):
    names = ('obs_map', 'obs_obj', 'reward', 'done')
    ta = (sa[0][0], sa[0][1], sa[1], sa[2])
    tb = (sb[0][0], sb[0][1], sb[1], sb[2])
    for nm, x, y in zip(names, ta, tb):
      if not torch.equal(x, y):
        diff = (x != y).reshape(B, -1).any(1).nonzero()[:, 0].tolist()
        print('  MISMATCH call %d (seed %d): %s differs in envs %s (shards %s)' % (t, seed, nm, diff[:16], sorted({i // G for i in diff})), flush=True)
        bad += 1
    if bad:
      describe('call %d' % t, a, shards, G)
      break
    act = a.sample()
    om, oo, r, d = shard_outputs()
    waits = []
    for k, s in enumerate(shards):
      if t % 2:                                   # staggered: current-stream work between the shards' launches
        filler = filler @ filler * 1e-3
      waits.append(s.step(act[k * G:(k + 1) * G], block=False, out=(om[k * G:(k + 1) * G], oo[k * G:(k + 1) * G], r[k * G:(k + 1) * G], d[k * G:(k + 1) * G])))
    sa = a.step(act)()
    for w in waits:
      w()
    sb = ((om, oo), r, d.view(torch.bool))
  print('compare B=%d L=%d shards=%d seed=%d: %s after %d calls' % (B, L, K, seed, 'MISMATCH' if bad else 'identical', t + 1), flush=True)
  a.close()
  for s in shards:
    s.close()
  return bad


def main():
  from stackrl_amd import assets as A
  pool = A.MeshPool.load(os.path.join(ROOT, 'tests', 'golden', 'ref_rocks.npz'))
  if os.environ.get('HEAVY', '1') != '0':
    heavy_prelude()
  total = 0
  for seed in (5, 6, 7):
    total += compare(pool, 64, 3, 4, seed, 3 + 2 + 4)
  total += compare(pool, 256, 8, 4, 11, 12)
  total += compare(pool, 128, 16, 2, 3, 20)
  print('diag_handles: %d mismatching comparisons' % total)


if __name__ == '__main__':
  main()
