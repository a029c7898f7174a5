// Hypothesis test for DESIGN.md section 6a: does a packed-fp32 instruction whose 64-bit source operand was assembled from
// TWO DIFFERENT earlier VALU instructions (lo half by one, hi half by another — what the SLP vectoriser produces when it
// packs scalar code) read a stale half when wavefronts of an MFMA kernel share the SIMD?  Fixed registers, no scheduling
// freedom: v10 <- v_mul, v11 <- v_add, then v_pk_fma reads v[10:11] at once; compared with the scalar evaluation.
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o pk_partial.so pk_partial.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void __launch_bounds__(128, 2) k_pk_partial(int iters, int gap, uint32_t* bad, float* sink) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a0 = 1.0f + 1e-3f * (float)(t & 1023), a1 = 0.5f + 2e-3f * (float)(t & 511);
  const float b0 = 0.999f, b1 = 1.001f;
  uint32_t n = 0;
  for (int i = 0; i < iters; ++i) {
    float r0, r1;
    // packed path: halves produced by two scalar instructions, consumed by one packed instruction
    if (gap == 0)
      asm volatile("v_mul_f32 v10, %2, %4\n v_add_f32 v11, %3, %5\n v_pk_fma_f32 v[12:13], v[10:11], v[10:11], v[10:11]\n"
                   "v_mov_b32 %0, v12\n v_mov_b32 %1, v13\n"
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13");
    else
      asm volatile("v_mul_f32 v10, %2, %4\n s_nop 0\n v_add_f32 v11, %3, %5\n s_nop 1\n v_pk_fma_f32 v[12:13], v[10:11], v[10:11], v[10:11]\n"
                   "v_mov_b32 %0, v12\n v_mov_b32 %1, v13\n"
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13");
    // scalar path
    float x = a0 * b0, y = a1 + b1;
    asm volatile("" : "+v"(x)); asm volatile("" : "+v"(y));
    float s0 = __builtin_fmaf(x, x, x), s1 = __builtin_fmaf(y, y, y);
    asm volatile("" : "+v"(s0)); asm volatile("" : "+v"(s1));
    n += (__float_as_uint(r0) != __float_as_uint(s0)) | (__float_as_uint(r1) != __float_as_uint(s1));
    a0 = 1.0f + (s0 - 1.0f) * 0.25f + 1e-6f * (float)(i & 15); a1 = 0.5f + (s1 - 0.5f) * 0.125f;
    if ((i & 63) == 63) { a0 = 1.0f + 1e-3f * (float)((t + i) & 1023); a1 = 0.5f + 2e-3f * (float)((t ^ i) & 511); }
  }
  if (n) atomicAdd(bad, n);
  if (a0 == 12345.0f) sink[0] = a0 + a1;
}
extern "C" int pk_partial(int iters, int blocks, int gap, void* stream, void* bad, void* sink) {
  hipLaunchKernelGGL(k_pk_partial, dim3(blocks), dim3(128), 0, (hipStream_t)stream, iters, gap, (uint32_t*)bad, (float*)sink);
  return (int)hipGetLastError();
}
