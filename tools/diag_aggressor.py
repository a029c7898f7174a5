#!/usr/bin/env python3
"""Which property of a co-resident kernel makes the packed-fp32 form of tools/experiments/pk_seq2.hip misread its operand?
The victim (3 launches of k_pk_seq2 on a side stream) runs beside one minimal aggressor at a time
(tools/experiments/pk_aggressor.hip), the aggressor relaunched on the current stream until the victim has finished.
Prints, per aggressor, the wrong low lanes of the failing forms (1: add, second source half-swapped; 10: multiply; 13 / 14: fma,
second source / addend) and of two clean forms (3 plain, 6 first source half-swapped) over the same number of executions.
Builds the two experiment libraries on the box (hipcc, a few seconds)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
here = os.path.join(ROOT, 'tools', 'experiments')
out = os.path.join(ROOT, 'gpurun_out')
os.makedirs(out, exist_ok=True)
for name in ('pk_seq2', 'pk_aggressor'):
  subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-fno-slp-vectorize', '-shared', '-fPIC', '-w',
                         '-o', os.path.join(out, name + '.so'), os.path.join(here, name + '.hip')])
V = ctypes.CDLL(os.path.join(out, 'pk_seq2.so'))
V.pk_seq2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
A = ctypes.CDLL(os.path.join(out, 'pk_aggressor.so'))
A.pk_aggressor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
NAMES = {-1: 'nothing', 0: 'vector FMAs', 1: 'MFMA bf16 16x16x32, accumulators in v registers', 2: 'the same MFMA, accumulators in a registers',
         3: 'MFMA f32 16x16x4, accumulators in v registers', 4: '1 + ds_read_b128', 5: 'DPP quad-permute moves', 6: 'v_perm_b32',
         7: 'SDWA adds', 8: '2 at s_setprio 3', 9: '1 at s_setprio 3', 10: 'MFMA bf16 32x32x16, accumulators in v registers',
         11: '1 with 160 accumulator registers per lane', 12: 'v_pk_fma_f32 with an SGPR-pair source',
         13: '1 with eight wait states after every MFMA', 14: 'MFMA f16 16x16x32', 15: 'MFMA bf16 16x16x16 (half-rate shape)', 16: 'MFMA i8 16x16x64',
         17: '1 with a vector-ALU write to the A operand every iteration', 18: '1 with a ds_read_b128 every iteration (result unused by the MFMAs)'}
kinds = [int(k) for k in sys.argv[1:]] or [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, -1]
blocks = int(os.environ.get('AGGRESSOR_BLOCKS', 2048))     # fewer blocks than CUs: does the victim fail on the other CUs too?
bad = torch.zeros(24, dtype=torch.int32, device='cuda'); sink = torch.zeros(16, device='cuda')
side = torch.cuda.Stream()
cur = torch.cuda.current_stream()
iters_v = int(os.environ.get('VICTIM_ITERS', 20000))
for k in kinds:
  bad.zero_(); torch.cuda.synchronize()
  side.wait_stream(cur)
  done = torch.cuda.Event()
  for _ in range(3):
    V.pk_seq2(iters_v, 2048, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(sink.data_ptr()))
  done.record(side)
  n = 0
  while not done.query():
    if k >= 0:
      A.pk_aggressor(k, 2000 * max(1, 2048 // blocks // 4), blocks, ctypes.c_void_p(cur.cuda_stream), ctypes.c_void_p(sink.data_ptr()))
      n += 1
      if n % 8 == 0:
        cur.synchronize()
  torch.cuda.synchronize()
  b = bad.tolist()
  ex = 3 * iters_v * 2048 * 128
  print('beside %-58s (%3d launches): wrong low lanes of %.1e executions: add %d  mul %d  fma src1 %d  fma addend %d | clean forms: plain %d  first source %d'
        % (NAMES[k], n, ex, b[1], b[10], b[16], b[17], b[3], b[6]), flush=True)
