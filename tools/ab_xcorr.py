#!/usr/bin/env python3
"""Which resource bounds the cross-correlation's forward (csrc/xcorr_mfma.hip, 128 x 128 x 16 map, 32 x 32 kernel, bf16x3)?
Ablation by source edit, as tools/ab_fused.py:

  ab_xcorr.py build     (here, CPU)  variants of xcorr_mfma.hip -> ab_libs/xcorr_*.so
  ab_xcorr.py run [B]   (GPU box)    times srl_xcorr_mfma(mode 0, precision 1) per variant

The variants' outputs are wrong by construction; only their durations mean anything."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, 'ab_libs')
SRC = os.path.join(ROOT, 'stackrl_amd', 'csrc', 'xcorr_mfma.hip')
A_READ = ('        af[u] = *(const bf16x8*)(ar + 16 * u);\n        if (SPLIT) al[u] = *(const bf16x8*)(ar + TILE + 16 * u);',
          '        af[u] = *(const bf16x8*)(ar + 16 * (u & 1));\n        if (SPLIT) al[u] = af[u];')
A_ONE = ('        af[u] = *(const bf16x8*)(ar + 16 * u);\n        if (SPLIT) al[u] = *(const bf16x8*)(ar + TILE + 16 * u);',
         '        af[u] = u < 2 ? *(const bf16x8*)(ar + 16 * u) : af[u - 2];\n        if (SPLIT) al[u] = u < 2 ? *(const bf16x8*)(ar + TILE + 16 * u) : al[u - 2];')
T_READ = ('        nf[j] = toeplitz_frag(ks + in * G::KR, j, lane);\n        if (SPLIT) nl[j] = toeplitz_frag(ks + KTILE + in * G::KR, j, lane);',
          '        nf[j] = tf[j];\n        if (SPLIT) nl[j] = tl[j];')
MFMA3 = ('''          if (SPLIT) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t + 2 * j], tl[j], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t + 2 * j], tf[j], acc[t], 0, 0, 0);
          }''', '''          if (SPLIT) { acc[t][0] += (float)tl[j][0] + (float)al[t + 2 * j][1]; }''')
T_MEMCPY = ('''  const int base = 16 + 32 * j + 8 * (lane >> 4) - (lane & 15);
  const uint32_t* w = (const uint32_t*)krow + (base >> 1);
  const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
  const uint32_t sh = (uint32_t)(base & 1) * 2u;
  union { uint32_t u[4]; bf16x8 v; } r;
  r.u[0] = __builtin_amdgcn_alignbyte(w1, w0, sh); r.u[1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
  r.u[2] = __builtin_amdgcn_alignbyte(w3, w2, sh); r.u[3] = __builtin_amdgcn_alignbyte(w4, w3, sh);
  return r.v;''', '''  const int base = 16 + 32 * j + 8 * (lane >> 4) - (lane & 15);
  bf16x8 r;
  __builtin_memcpy(&r, krow + base, 16);
  return r;''')
VARIANTS = {
  'default': [],
  'toeplitz_unaligned_reads': [T_MEMCPY],
  'a_reads_2_of_9': [A_ONE],        # two windows read per plane, the rest copied: LDS traffic of the A fragments / 4.5
  'no_toeplitz_reads': [T_READ],
  'one_mfma_of_3': [MFMA3],
  'a2_and_no_toeplitz': [A_ONE, T_READ],
}


def build():
  os.makedirs(AB, exist_ok=True)
  src = open(SRC).read()
  for name, edits in VARIANTS.items():
    s = src
    for old, new in edits:
      assert old in s, (name, old[:60])
      s = s.replace(old, new)
    path = os.path.join(os.path.dirname(SRC), '_abx_%s.hip' % name)
    open(path, 'w').write(s)
    try:
      subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fno-slp-vectorize', '-w',
                             path, '-o', os.path.join(AB, 'xcorr_%s.so' % name)])
    finally:
      os.remove(path)
    print('built', name, flush=True)


def run(B):
  sys.path.insert(0, ROOT)
  import torch
  g = torch.Generator(device='cuda').manual_seed(1)
  x = torch.randn((B, 16, 128, 128), generator=g, device='cuda')
  w = torch.randn((B, 16, 32, 32), generator=g, device='cuda')
  out = torch.empty((B, 97, 97), device='cuda')
  VP, I = ctypes.c_void_p, ctypes.c_int32
  st = VP(torch.cuda.current_stream().cuda_stream)
  for name in VARIANTS:
    L = ctypes.CDLL(os.path.join(AB, 'xcorr_%s.so' % name))
    f = L.srl_xcorr_mfma
    f.argtypes = [I, I, VP, I, VP, I, VP, VP, ctypes.c_int64, I, I, I, I, VP]
    call = lambda: f(0, 1, x.data_ptr(), 1, w.data_ptr(), 1, out.data_ptr(), None, 0, B, 16, 128, 32, st)
    for _ in range(3):
      assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
      call()
    e1.record(); torch.cuda.synchronize()
    print('%-22s %8.1f us per launch (B = %d)' % (name, e0.elapsed_time(e1) * 100.0, B), flush=True)


if __name__ == '__main__':
  if sys.argv[1] == 'build':
    build()
  else:
    run(int(sys.argv[2]) if len(sys.argv) > 2 else 512)
