#!/usr/bin/env python3
"""Rollout policy time per 4,096 samples by chunk size (qops.FusedPolicy)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import nets, qops
B = 4096
net = nets.DeepQSiamFCN(seed=1).cuda().eval()
g = torch.Generator(device='cuda').manual_seed(0)
xm = torch.randint(0, 256, (B, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8)
xo = torch.randint(0, 256, (B, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8)
for dt in (torch.bfloat16, None):
  for chunk in (256, 512, 768, 1024, 2048):
    pol = qops.FusedPolicy(chunk=chunk, autocast=dt, fast=True)
    with torch.no_grad():
      for _ in range(2): pol(net, (xm, xo), 0.1, g)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(3): pol(net, (xm, xo), 0.1, g)
      torch.cuda.synchronize()
    print('%s chunk %4d: %.2f ms per 4,096 samples' % ('bf16' if dt else 'fp32-class', chunk, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    del pol
    torch.cuda.empty_cache()
