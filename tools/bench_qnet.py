#!/usr/bin/env python3
"""Rollout forward of the Q-net: dtype / memory-format / batch-chunk variants (library convs via MIOpen)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import nets, qops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
net = nets.DeepQSiamFCN(seed=1).cuda().eval()
g = torch.Generator(device='cuda').manual_seed(0)
xm = torch.randint(0, 256, (B, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8)
xo = torch.randint(0, 256, (B, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8)
macs = sum(nets.forward_macs().values())
def run(tag, chunk, autocast, cl, fast=False):
  n = net.to(memory_format=torch.channels_last) if cl else net.to(memory_format=torch.contiguous_format)
  pol = qops.FusedPolicy(chunk=chunk, autocast=autocast, fast=fast)
  with torch.no_grad():
    for _ in range(2): pol(n, (xm, xo), 0.1, g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): pol(n, (xm, xo), 0.1, g)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
  print('%-40s %.1f ms  %.1f TFLOP/s' % (tag, dt * 1e3, 2 * macs * B / dt / 1e12), flush=True)
run('fp32 nchw chunk 512', 512, None, False)
run('fp32 nchw chunk 128', 128, None, False)
run('fp32 channels_last chunk 512', 512, None, True)
run('bf16 nchw chunk 512', 512, torch.bfloat16, False)
run('bf16 channels_last chunk 512', 512, torch.bfloat16, True)
run('fp16 channels_last chunk 512', 512, torch.float16, True)
run('bf16 fused epilogues chunk 512', 512, torch.bfloat16, False, True)
run('bf16 fused epilogues chunk 1024', 1024, torch.bfloat16, False, True)
_ff = qops.FastFeatures
qops.FastFeatures = lambda net: _ff(net, mfma_conv=False)
run('bf16 fused epilogues, library convs only', 1024, torch.bfloat16, False, True)
qops.FastFeatures = _ff

# the cross-correlation alone: fp32 vector kernel vs the bf16 MFMA kernel
from stackrl_amd import qops as _q
xf = torch.rand((B, 16, 128, 128), device='cuda'); wf = torch.rand((B, 16, 32, 32), device='cuda') - 0.3
for tag, a, b, pr in (('xcorr fp32 VALU', xf, wf, 'fp32'), ('xcorr bf16x3 MFMA (fp32 operands)', xf, wf, None), ('xcorr bf16 MFMA', xf.to(torch.bfloat16), wf.to(torch.bfloat16), None)):
  for _ in range(2): _q.xcorr_forward(a, b, pr)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(5): _q.xcorr_forward(a, b, pr)
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
  useful = 2.0 * 97 * 97 * 32 * 32 * 16 * B
  print('%-40s %.2f ms  %.1f useful TFLOP/s' % (tag, dt * 1e3, useful / dt / 1e12), flush=True)
