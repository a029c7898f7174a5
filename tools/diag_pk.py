#!/usr/bin/env python3
"""Packed-fp32 instructions under co-resident kernels (tools/experiments/pk_victim.hip): mismatches between the packed and
the scalar evaluation of one recurrence, alone and while the Q-net's convolution kernels run on another stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stackrl_amd import nets, qops
here = os.path.dirname(os.path.abspath(__file__))
V = ctypes.CDLL(os.path.join(here, 'experiments', 'pk_victim.so'))
V.pk_victim.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
bad = torch.zeros(1, dtype=torch.int32, device='cuda'); sink = torch.zeros(4, device='cuda')
side = torch.cuda.Stream()
net = nets.DeepQSiamFCN(seed=2).cuda()
ff = qops.FastFeatures(net, dtype=torch.float32)
C = torch.randn(256, 1, 97, 97, device='cuda')
xm = torch.randint(0, 256, (256, 128, 128, 2), device='cuda', dtype=torch.uint8)
xo = torch.randint(0, 256, (256, 32, 32, 1), device='cuda', dtype=torch.uint8)
filler = torch.randn(2048, 2048, device='cuda')
with torch.no_grad():
  ff.pos(C); ff((xm, xo))
torch.cuda.synchronize()
PP = ctypes.CDLL(os.path.join(here, 'experiments', 'pk_partial.so'))
PP.pk_partial.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
for gap in (0, 1):
  for load in ('none', 'pos', 'ff', 'none'):
    bad.zero_(); torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    for _ in range(3):
      PP.pk_partial(100000, 2048, gap, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(sink.data_ptr()))
    with torch.no_grad():
      for _ in range(40):
        if load == 'pos': ff.pos(C)
        elif load == 'ff': ff((xm, xo))
    torch.cuda.synchronize()
    print('halves from two scalar instructions, gap', gap, 'load', load, ': packed results that differ from the scalar ones:', int(bad.item()), flush=True)
if os.environ.get('ONLY_PARTIAL'): sys.exit(0)
V.pk_victim_dead.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
for junk in (0x00000001, 0x7fa00000, 0x7f800000, 0xffffffff, 0x00400000, 0x80000001):
  for load in ('none', 'pos', 'ff'):
    bad.zero_(); torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    for _ in range(3):
      V.pk_victim_dead(100000, 2048, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(sink.data_ptr()), junk)
    with torch.no_grad():
      for _ in range(40):
        if load == 'pos': ff.pos(C)
        elif load == 'ff': ff((xm, xo))
    torch.cuda.synchronize()
    print('dead-half pattern', hex(junk), 'load', load, ': live-half results that differ from the scalar ones:', int(bad.item()), flush=True)
for load in ('none', 'matmul', 'pos', 'ff', 'none'):
  bad.zero_(); torch.cuda.synchronize()
  side.wait_stream(torch.cuda.current_stream())
  for _ in range(4):
    V.pk_victim(200000, 2048, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(sink.data_ptr()))
  with torch.no_grad():
    for _ in range(60):
      if load == 'pos': ff.pos(C)
      elif load == 'ff': ff((xm, xo))
      elif load == 'matmul': filler = (filler @ filler) * 1e-3
  torch.cuda.synchronize()
  print('load', load, ': iterations in which the packed and the scalar results differ:', int(bad.item()), flush=True)
