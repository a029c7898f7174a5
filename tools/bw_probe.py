#!/usr/bin/env python3
"""HBM probes on the box: pure-write (fill), copy and read-reduce rates at the render kernel's footprint."""
import torch, time
def timeit(f, n=30):
  for _ in range(5): f()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): f()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e-3
for mb in (106, 512, 2048):
  n = mb * 1024 * 1024 // 4
  a = torch.empty(n, dtype=torch.float32, device='cuda'); b = torch.empty_like(a)
  t = timeit(lambda: a.fill_(1.0)); print('fill   %5d MB: %.2f TB/s (%.1f us)' % (mb, n * 4 / t / 1e12, t * 1e6))
  t = timeit(lambda: b.copy_(a));   print('copy   %5d MB: %.2f TB/s r+w' % (mb, 2 * n * 4 / t / 1e12))
  t = timeit(lambda: a.sum());      print('reduce %5d MB: %.2f TB/s' % (mb, n * 4 / t / 1e12), flush=True)
