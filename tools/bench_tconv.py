#!/usr/bin/env python3
"""Per-layer timing of the update path's convolution kernels (csrc/train_conv.hip) at the shapes of `DeepQSiamFCN`:
forward / data gradient (`srl_tconv`) and weight gradient (`srl_twrw`) against the float32 MFMA peak (157.3 TFLOP/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import qtrain

PEAK = 157.3e12
LAYERS = [(2, 16, 128), (16, 16, 128), (32, 16, 128), (16, 32, 64), (32, 32, 64), (64, 32, 64), (32, 64, 32), (64, 64, 32), (128, 64, 32),
          (64, 128, 16), (128, 128, 16), (256, 128, 16), (128, 256, 8), (256, 256, 8), (1, 16, 97)]


def timeit(f, n=20):
  for _ in range(3): f()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): f()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n * 1e3


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
  sc = qtrain._Scratch()
  print('B = %d' % B)
  for cin, cout, H in LAYERS:
    net = torch.nn.Sequential(torch.nn.Conv2d(cin, cout, 3, padding=1)).cuda()
    P = qtrain.Packed(net); P.refresh()
    conv = net[0]
    x = qtrain.Act(torch.randn(B, H, H, cin, device='cuda'))
    gz = torch.randn(B, H, H, cout, device='cuda')
    gw = torch.zeros_like(conv.weight)
    flops = 2.0 * B * H * H * cin * cout * 9
    tf = timeit(lambda: qtrain.tconv(x, P.w(conv, 0), conv.bias, cout))
    tw = timeit(lambda: qtrain.twrw(x, gz, gw, sc))
    line = '%3d -> %3d @ %3d^2: fwd %7.1f us (%4.1f %% of fp32 MFMA peak)   wrw %7.1f us (%4.1f %%)' % (
      cin, cout, H, tf, 100 * flops / (tf * 1e-6) / PEAK, tw, 100 * flops / (tw * 1e-6) / PEAK)
    if cin >= 16:
      td = timeit(lambda: qtrain.tconv(qtrain.Act(gz), P.w(conv, 1), None, cin, relu=False))
      line += '   dgrad %7.1f us (%4.1f %%)' % (td, 100 * flops / (td * 1e-6) / PEAK)
    print(line, flush=True)


if __name__ == '__main__':
  main()
