#!/usr/bin/env python3
"""Experiment: the rollout forward of 4,096 samples as two chunks of 2,048 in sequence on one stream (the product) against the
two chunks on two streams at once (one's kernel tails under the other's kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import nets, qops

B = 4096
net = nets.DeepQSiamFCN(seed=1).cuda().eval()
g = torch.Generator(device='cuda').manual_seed(3)
xm = torch.randint(0, 256, (B, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8)
xo = torch.randint(0, 256, (B, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8)
gen = torch.Generator(device='cuda').manual_seed(5)
pol = qops.FusedPolicy(chunk=2048, autocast=None, fast=True)
pa, pb = qops.FusedPolicy(chunk=2048, autocast=None, fast=True), qops.FusedPolicy(chunk=2048, autocast=None, fast=True)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def seq():
  return pol(net, (xm, xo), 0.0, gen)


def par():
  cur = torch.cuda.current_stream()
  sa.wait_stream(cur); sb.wait_stream(cur)
  with torch.cuda.stream(sa):
    a = pa(net, (xm[:2048], xo[:2048]), 0.0, gen)
  with torch.cuda.stream(sb):
    b = pb(net, (xm[2048:], xo[2048:]), 0.0, gen)
  cur.wait_stream(sa); cur.wait_stream(sb)
  return a, b


for name, f in (('one stream ', seq), ('two streams', par), ('one stream ', seq), ('two streams', par)):
  for _ in range(2): f()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(5): f()
  torch.cuda.synchronize()
  print('%s: %.2f ms per 4,096 samples' % (name, 1e3 * (time.perf_counter() - t0) / 5))
