#!/usr/bin/env python3
"""The rollout's precision against the reference's, with a number (dqn.py:330-348: the greedy action is
the arg-max of Q over the 9,409 placements, computed by the reference in float32, models.py:144-147).

On N real observations (random-policy Stack-v0 episodes of this repo's env) and a He-initialised DeepQSiamFCN, the advantages
(the arg-max of Q = A - mean(A) + V is the arg-max of A, models.py:179-192) are evaluated
  (64)     in float64 by the module graph (the yardstick),
  (f32)    in float32 by the stock module graph (library convolutions: the reference's dtype),
  (bf16x3) by the fp32-class rollout bench.py times as `dqn.bf16x3` (FastFeatures(float32) + k_xcorr_mfma + fused position head),
  (bf16)   by the bf16 rollout (labelled narrower),
and for each of the three the tool reports: max |A - A64| / range(A64) over the samples, the share of samples whose arg-max
differs from float64's (`flip_rate`), and the largest top-2 gap of float64 (relative to the range) at which a flip occurred —
a path is "arg-max-exact above gap g" when no sample with a float64 gap above g flips.

usage: python tools/argmax_precision.py [N=4608] [seed=1]  ->  one JSON object on stdout"""
import copy
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch


def observations(n, seed=3, B=512, L=8):
  """n real observations: the states a random policy visits (reset observation included), whole episodes."""
  from stackrl_amd import assets, env as envs
  g = envs.VecStackEnv(n_parallel=B, seed=seed, pool=assets.default_pool(), block=True, episode_length=L)
  maps, objs = [], []
  (m, o), _, _ = g.reset()
  while sum(x.shape[0] for x in maps) < n:
    keep = (o.flatten(1).amax(1) > 0)             # a rock is on show (the terminal observation of an episode shows none)
    maps.append(m[keep].clone()); objs.append(o[keep].clone())
    (m, o), _, _ = g.step(g.sample())
  g.close()
  return torch.cat(maps)[:n], torch.cat(objs)[:n]


@torch.no_grad()
def advantages(net, xm, xo, how, chunk=128):
  from stackrl_amd import nets, qops
  out = []
  ff = None
  if how == '64':
    net = copy.deepcopy(net).double()
  for s in range(0, xm.shape[0], chunk):
    a, b = xm[s:s + chunk], xo[s:s + chunk]
    if how == '64':
      x, w = a.permute(0, 3, 1, 2).double() / 255.0, b.permute(0, 3, 1, 2).double() / 255.0      # models.py:144-147
      x, _ = net.left(x); w, _ = net.right(w)
      adv = net.pos(nets.correlation_reference(x, w)).flatten(1)
    elif how == 'f32':
      x, _, w = net.features((a, b))
      adv = net.pos(nets.correlation_reference(x, w)).flatten(1)
    else:
      if ff is None:
        ff = qops.FastFeatures(net, dtype=torch.float32 if how == 'bf16x3' else torch.bfloat16)
      x, w = ff((a, b))
      adv = ff.pos(qops.xcorr_forward(x, w))
    out.append(adv.double())
  return torch.cat(out)


def compare(a64, a):
  rng = (a64.amax(1) - a64.amin(1))
  err = ((a - a64).abs().amax(1) / rng)
  top2 = a64.topk(2, dim=1).values
  gap = (top2[:, 0] - top2[:, 1]) / rng
  flip = a.argmax(1) != a64.argmax(1)
  # what a flip costs: the float64 advantage given up, relative to the range
  regret = (top2[:, 0] - a64.gather(1, a.argmax(1)[:, None])[:, 0]) / rng
  return {
    'max_rel_err': float(err.max()), 'mean_rel_err': float(err.mean()),
    'flip_rate': float(flip.double().mean()), 'flips': int(flip.sum()),
    'largest_gap_with_a_flip': float(gap[flip].max()) if bool(flip.any()) else 0.0,
    'max_regret_of_a_flip': float(regret.max()),
  }, gap


def run(n=4608, seed=1):
  from stackrl_amd import nets
  xm, xo = observations(n)
  net = nets.DeepQSiamFCN(seed=seed).cuda().eval()
  a64 = advantages(net, xm, xo, '64')
  out = {'samples': int(xm.shape[0]), 'net_seed': seed, 'observations': 'random-policy Stack-v0 episodes (8 rocks), states with a rock on show',
         'actions': int(a64.shape[1])}
  gap = None
  for how in ('f32', 'bf16x3', 'bf16'):
    out[how], gap = compare(a64, advantages(net, xm, xo, how))
  q = torch.tensor([0.01, 0.1, 0.5], dtype=torch.float64, device=gap.device)
  out['float64_top2_gap_over_range'] = {'min': float(gap.min()), 'p1': float(gap.quantile(q[0])), 'p10': float(gap.quantile(q[1])),
                                        'median': float(gap.quantile(q[2]))}
  for g in (1e-6, 1e-5, 1e-4, 1e-3):
    out['share_of_samples_with_gap_below_%g' % g] = float((gap < g).double().mean())
  return out


if __name__ == '__main__':
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
  seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
  print(json.dumps(run(n, seed), indent=1))
