#!/usr/bin/env python3
"""The rollout's cross-correlation forward (16 x 128^2 map, 16 x 32^2 kernel per sample), Toeplitz kernel against the
row-product kernel (SRL_XCORR_ROWS=0 / 1), float32 operands with the bf16x3 split and bf16 operands: us per launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import qops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.rand((B, 16, 128, 128), generator=g, device='cuda')
w = torch.rand((B, 16, 32, 32), generator=g, device='cuda') - 0.3
for name, a, k, prec in (('bf16x3', x, w, qops.BF16X3), ('bf16', x.to(torch.bfloat16), w.to(torch.bfloat16), qops.BF16)):
  outs = {}
  for rows in ('0', '1', '8'):
    os.environ['SRL_XCORR_ROWS'] = '0' if rows == '0' else '1'
    os.environ['SRL_XCORR_ROWS_WAVES'] = '8' if rows == '8' else '4'
    for _ in range(3):
      o = qops.xcorr_forward_mfma(a, k, prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      o = qops.xcorr_forward_mfma(a, k, prec)
    e1.record()
    torch.cuda.synchronize()
    outs[rows] = o
    print('%s B=%d %s: %.1f us per launch' % (name, B, {'0': 'Toeplitz            ', '1': 'row product, 4 waves', '8': 'row product, 8 waves'}[rows], 1e3 * e0.elapsed_time(e1) / 20))
  print('  max |rows - toeplitz| / scale = %.2e' % (float((outs['0'] - outs['1']).abs().max()) / float(outs['0'].abs().max())))
