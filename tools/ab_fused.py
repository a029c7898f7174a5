#!/usr/bin/env python3
"""Which phase bounds the fused thin + 16 -> 16 kernel (csrc/conv_mfma.hip k_thin_conv3x3_x3)?  Ablation by source edit:

  ab_fused.py build     (here, CPU: hipcc cross-compiles)  variants of conv_mfma.hip with one phase removed (or, the last two,
                        with plain stores / the compiler's two workgroups per CU: what the kernel had before) -> ab_libs/fused_*.so
  ab_fused.py run [B]   (GPU box)  times srl_thin_conv3x3_bias_relu_f32 (uint8 128 x 128 x 2 -> skip slice + pooled) per variant

The variants' outputs are wrong by construction; only their durations mean anything."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, 'ab_libs')
SRC = os.path.join(ROOT, 'stackrl_amd', 'csrc', 'conv_mfma.hip')
LB2 = ('template <int CT, typename TIN, bool PROJ>\n__global__ void __launch_bounds__(256, 4)', 'template <int CT, typename TIN, bool PROJ>\n__global__ void __launch_bounds__(256, 2)')
NT_OFF = ('  __builtin_nontemporal_store(v, (f32x4*)p);', '  *(f32x4*)p = v;')
VARIANTS = {
  'default': [],
  'no_thin_fma': [('      thin_outputs<CT, 8>(v, w1 + half * 8 * CT * 9, b1 + half * 8, r);',
                   '      for (int i = 0; i < 8; ++i) r[i] = v[i % (9 * CT)] + b1[half];')],
  'no_mfma': [('  x3_mfma_pass<16, RW, 1>(tile, PLANE, row0, n, g, wh, wl, acc);\n',
               '  for (int r = 0; r < RW; ++r) acc[r][0][0] += (float)tile[(row0 + r) * G::TW * G::PS + n] + (float)wh[r][0][0] + (float)wl[r][0][1];\n')],
  'one_round': [('  for (int j = 0; j < 3; ++j) {\n    const int wid = __builtin_amdgcn_readfirstlane(4 * j + wave);   // 0 .. 11',
                 '  for (int j = 0; j < 1; ++j) {\n    const int wid = __builtin_amdgcn_readfirstlane(4 * j + wave);   // 0 .. 11')],
  'no_store': [('  x3_epilogue<16, 1, RW, PROJ>(acc, bias, out, pooled, H, W, ostride, ooff, nchw, pw, pb, proj_out, H, W, b, x0, y0, 0, row0, n, g);\n}',
                '  if (acc[0][0][0] == 12345.678f) x3_epilogue<16, 1, RW, PROJ>(acc, bias, out, pooled, H, W, ostride, ooff, nchw, pw, pb, proj_out, H, W, b, x0, y0, 0, row0, n, g);\n}')],
  'plain_stores': [NT_OFF],
  'lb2': [LB2],
}


def build():
  os.makedirs(AB, exist_ok=True)
  src = open(SRC).read()
  for name, edits in VARIANTS.items():
    s = src
    for old, new in edits:
      assert old in s, (name, old[:60])
      s = s.replace(old, new)
    path = os.path.join(os.path.dirname(SRC), '_ab_%s.hip' % name)
    open(path, 'w').write(s)
    try:
      subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fno-slp-vectorize', '-w',
                             path, '-o', os.path.join(AB, 'fused_%s.so' % name)])
    finally:
      os.remove(path)
    print('built', name, flush=True)


def run(B):
  sys.path.insert(0, ROOT)
  import torch
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(1)
  x = torch.randint(0, 256, (B, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8)
  w1 = qops.pack_thin_weights((torch.rand((16, 2, 3, 3), generator=g, device='cuda') - 0.5) * 0.5)
  b1 = torch.rand(16, generator=g, device='cuda') - 0.5
  wf = qops.pack_conv3x3_weights_x3((torch.rand((16, 16, 3, 3), generator=g, device='cuda') - 0.5) * 0.2)
  b2 = torch.rand(16, generator=g, device='cuda') - 0.5
  cat = torch.zeros((B, 32, 128, 128), device='cuda').contiguous(memory_format=torch.channels_last)
  pooled = torch.zeros((B, 16, 64, 64), device='cuda').contiguous(memory_format=torch.channels_last)
  VP, I = ctypes.c_void_p, ctypes.c_int32
  st = VP(torch.cuda.current_stream().cuda_stream)
  for name in VARIANTS:
    L = ctypes.CDLL(os.path.join(AB, 'fused_%s.so' % name))
    f = L.srl_thin_conv3x3_bias_relu_f32
    f.argtypes = [VP, I, I] + [VP] * 6 + [I] * 6 + [VP]
    call = lambda: f(x.data_ptr(), 0, 2, w1.data_ptr(), b1.data_ptr(), wf.data_ptr(), b2.data_ptr(), cat.data_ptr(), pooled.data_ptr(),
                     B, 128, 128, 32, 16, 0, st)
    for _ in range(3):
      assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
      call()
    e1.record(); torch.cuda.synchronize()
    print('%-14s %8.1f us per launch (B = %d)' % (name, e0.elapsed_time(e1) * 100.0, B), flush=True)


if __name__ == '__main__':
  if sys.argv[1] == 'build':
    build()
  else:
    run(int(sys.argv[2]) if len(sys.argv) > 2 else 512)
