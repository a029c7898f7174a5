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
PS2 = ctypes.CDLL(os.path.join(here, 'experiments', 'pk_seq2.so'))
PS2.pk_seq2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
bad8 = torch.zeros(24, dtype=torch.int32, device='cuda'); sink = torch.zeros(16, device='cuda')
for load in ('none', 'ff', 'none'):
  bad8.zero_(); torch.cuda.synchronize()
  side.wait_stream(torch.cuda.current_stream())
  for _ in range(3):
    PS2.pk_seq2(40000, 2048, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(bad8.data_ptr()), ctypes.c_void_p(sink.data_ptr()))
  with torch.no_grad():
    for _ in range(60):
      if load == 'ff': ff((xm, xo))
  torch.cuda.synchronize()
  print('consumer [0 swap+neg, 1 swap, 2 neg, 3 plain, 4 swap+neg after 32-bit producers, 5 swap+neg after packed producers with VGPR sources, 6 swap of src0, 7 low half of src1 for both, 8 high half of src1 for both, 9 fma high half of src0 for both, 10 mul swap] load', load, ': mismatches', bad8[:11].tolist(), '; of form 1: low lane = the UNSWAPPED sum', int(bad8[11]), ', high lane wrong', int(bad8[12]), '; SGPR pair with its high half for the low lane: as FIRST source', int(bad8[13]), ', as SECOND source', int(bad8[14]), '; packed fma with the second source half-swapped', int(bad8[16]), ', with the addend half-swapped', int(bad8[17]), flush=True)
  if int(bad8[15]):
    x = sink[8:15].tolist(); print('   a failing case of form 1: v[0:1] =', x[0:2], 'v[32:33] =', x[2:4], 'packed result (lo, hi) =', x[4:6], 'expected lo = v0 + v33 =', x[6], '; v0 + v32 =', x[0] + x[2], flush=True)
if os.environ.get('ONLY_SEQ2'): sys.exit(0)
PS = ctypes.CDLL(os.path.join(here, 'experiments', 'pk_seq.so'))
PS.pk_seq.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
bad4 = torch.zeros(8, dtype=torch.int32, device='cuda')
net_ = net
dbuf_ = torch.zeros((256, 16, 112, 112), dtype=torch.float32, device='cuda').contiguous(memory_format=torch.channels_last)
for load in ('none', 'proj', 'ff', 'none'):
  bad4.zero_(); torch.cuda.synchronize()
  side.wait_stream(torch.cuda.current_stream())
  for _ in range(3):
    PS.pk_seq(60000, 2048, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(bad4.data_ptr()), ctypes.c_void_p(sink.data_ptr()))
  with torch.no_grad():
    for _ in range(60):
      if load == 'proj': qops.conv3x3_relu_project(dbuf_, ff._wf[net_.pos[2]], ff._w[net_.pos[2]][1], ff._pos[0], ff._pos[1], 97, 97)
      elif load == 'ff': ff((xm, xo))
  torch.cuda.synchronize()
  print('packed producers (SGPR-pair source) -> half-swapped packed consumers, s_nop between = none / 1 / 2 / 4: load', load, ': mismatches', bad4[:4].tolist(), '; back to back: packed consumer != value computed outside', int(bad4[4]), ', scalar consumer != it', int(bad4[5]), flush=True)
if os.environ.get('ONLY_SEQ'): sys.exit(0)
PF = ctypes.CDLL(os.path.join(here, 'experiments', 'pk_forms.so'))
PF.pk_forms.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
bad6 = torch.zeros(24, dtype=torch.int32, device='cuda'); sink = torch.zeros(16, device='cuda')
pw_, pb_ = net.pos[4].weight.detach().reshape(-1).contiguous(), net.pos[4].bias.detach()
dbuf = torch.zeros((256, 16, 112, 112), dtype=torch.float32, device='cuda').contiguous(memory_format=torch.channels_last)
for load in ('none', 'proj', 'ff', 'none'):
  bad6.zero_(); torch.cuda.synchronize()
  side.wait_stream(torch.cuda.current_stream())
  for _ in range(3):
    PF.pk_forms(60000, 2048, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(bad6.data_ptr()), ctypes.c_void_p(sink.data_ptr()))
  with torch.no_grad():
    for _ in range(60):
      if load == 'proj': qops.conv3x3_relu_project(dbuf, ff._wf[net.pos[2]], ff._w[net.pos[2]][1], ff._pos[0], ff._pos[1], 97, 97)
      elif load == 'ff': ff((xm, xo))
  torch.cuda.synchronize()
  print('packed forms [A add neg src1, B op_sel_hi, C inline 0.5, D op_sel fma, E sgpr pair + neg src0, F -x - 0, G add neg src0, H mul neg src1, I fma neg addend, J add plain, K scalar add neg, L add op_sel+op_sel_hi+neg, M add op_sel+neg, N0..N3 packed producer -> half-swapped packed consumer with 0..3 instructions between] load', load, ': mismatches', bad6[:17].tolist(), flush=True)
  if int(bad6[11]):
    x = sink[8:16].tolist(); print('   example of form L: a =', x[0:2], 'b =', x[2:4], 'returned', x[4:6], 'expected (a0 - b1, a1 - b0) =', x[6:8], flush=True)
if os.environ.get('ONLY_FORMS'): sys.exit(0)
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
