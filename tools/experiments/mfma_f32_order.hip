// Which arithmetic does v_mfma_f32_32x32x2_f32 perform?  D = C + A[:,0] B[0,:] + A[:,1] B[1,:] per element; compare
// against the candidate orders on random data.  (Experiment for an MFMA plane sweep in srl_k_render: usable only if the
// result equals a chain of IEEE fused multiply-adds that the CPU oracle can restate.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* a, const float* b, const float* c, float* d) {
  // A: 32 x 2 (lane l: row l % 32, k = l / 32), B: 2 x 32 (lane l: col l % 32, k = l / 32)
  const int l = threadIdx.x;
  const int blk = blockIdx.x;
  const float av = a[blk * 64 + l], bv = b[blk * 64 + l];
  f32x16 cv;
  for (int r = 0; r < 16; ++r) cv[r] = c[(blk * 64 + l) * 16 + r];
  f32x16 dv = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, cv, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[(blk * 64 + l) * 16 + r] = dv[r];
}
int main() {
  const int NB = 4096;
  size_t n = (size_t)NB * 64;
  float *a = (float*)malloc(n * 4), *b = (float*)malloc(n * 4), *c = (float*)malloc(n * 64), *d = (float*)malloc(n * 64);
  srand(3);
  auto rnd = []() { return (float)((rand() / (double)RAND_MAX - 0.5) * 2.0) * ((rand() & 3) == 0 ? 1e-3f : 1.0f); };
  for (size_t i = 0; i < n; ++i) { a[i] = rnd(); b[i] = rnd(); }
  for (size_t i = 0; i < n * 16; ++i) c[i] = rnd();
  float *da, *db, *dc, *dd;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 64); hipMalloc(&dd, n * 64);
  hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c, n * 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, da, db, dc, dd);
  hipMemcpy(d, dd, n * 64, hipMemcpyDeviceToHost);
  // D layout (32x32): lane l holds column j = l % 32, rows i = 8 * (r / 4) + 4 * (l / 32) + r % 4
  long long m01 = 0, m10 = 0, mfused = 0, msep = 0, tot = 0;
  for (int blk = 0; blk < NB; ++blk)
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int j = l % 32, i = 8 * (r / 4) + 4 * (l / 32) + r % 4;
        const float a0 = a[blk * 64 + i], a1 = a[blk * 64 + 32 + i], b0 = b[blk * 64 + j], b1 = b[blk * 64 + 32 + j];
        const float cc = c[((size_t)blk * 64 + l) * 16 + r], got = d[((size_t)blk * 64 + l) * 16 + r];
        const float o01 = fmaf(a1, b1, fmaf(a0, b0, cc));     // k = 0 first
        const float o10 = fmaf(a0, b0, fmaf(a1, b1, cc));     // k = 1 first
        const float of = (float)((double)a0 * b0 + (double)a1 * b1 + (double)cc);   // one rounding (nearly: double is exact enough)
        const float os = (a0 * b0 + a1 * b1) + cc;            // separate roundings
        tot++; m01 += got == o01; m10 += got == o10; mfused += got == of; msep += got == os;
      }
  printf("elements %lld: == fma(a1,b1,fma(a0,b0,c)) %lld | == fma(a0,b0,fma(a1,b1,c)) %lld | == single rounding %lld | == separate roundings %lld\n",
         tot, m01, m10, mfused, msep);
  return 0;
}
