import os, sys, time
sys.path.insert(0, os.getcwd())
from stackrl_amd import build as b
if len(sys.argv) > 1 and sys.argv[1] != 'product':
  b.QLIB = os.path.abspath(sys.argv[1]); b.qstale = lambda: False
import torch
from stackrl_amd import nets, qops
net = nets.DeepQSiamFCN(seed=2).cuda()
g = torch.Generator(device='cuda').manual_seed(1)
xm = torch.randint(0, 256, (2048, 128, 128, 2), device='cuda', dtype=torch.uint8, generator=g)
xo = torch.randint(0, 256, (2048, 32, 32, 1), device='cuda', dtype=torch.uint8, generator=g)
for dt in (None, torch.bfloat16):
  pol = qops.FusedPolicy(autocast=dt, fast=True)
  gen = torch.Generator(device='cuda').manual_seed(3)
  for _ in range(2): pol(net, (xm, xo), 1.0, gen)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(6): pol(net, (xm, xo), 1.0, gen)
  torch.cuda.synchronize(); print(sys.argv[1] if len(sys.argv) > 1 else 'product', 'bf16' if dt else 'bf16x3', 'forward of 2048 samples: %.2f ms' % ((time.perf_counter() - t0) / 6 * 1e3), flush=True)
