#!/usr/bin/env python3
"""k_conv3x3_gemm against the library convolution + fused bias pass, per deep U-Net layer (512 samples)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import qops
LAYERS = [(32, 64, 32), (64, 64, 32), (128, 64, 32), (64, 128, 16), (128, 128, 16), (256, 128, 16), (128, 256, 8), (256, 256, 8)]
B = 512


def timeit(fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / reps * 1e3


for dt in (torch.bfloat16, torch.float32):
  tot = [0.0, 0.0]
  for cin, cout, W in LAYERS:
    x = torch.randn(B, cin, W, W, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    wl = w.to(dt).contiguous(memory_format=torch.channels_last)
    wf = qops.pack_conv3x3_gemm_weights(w, x3=(dt == torch.float32))
    lib = timeit(lambda: qops.bias_act(torch.nn.functional.conv2d(x, wl, None, padding=1), b))
    mine = timeit(lambda: qops.conv3x3_gemm_bias_relu(x, wf, b, cout))
    gf = 2.0 * B * W * W * cin * cout * 9 / 1e6     # MFLOP: MFLOP / us = TFLOP/s
    tot[0] += lib; tot[1] += mine
    print('%s %3d -> %3d @ %2d^2: library + bias pass %7.1f us (%5.0f TFLOP/s)   k_conv3x3_gemm %7.1f us (%5.0f TFLOP/s)' % (
      str(dt)[6:], cin, cout, W, lib, gf / lib, mine, gf / mine), flush=True)
  print('%s total: library %.0f us, k_conv3x3_gemm %.0f us' % (str(dt)[6:], tot[0], tot[1]))
