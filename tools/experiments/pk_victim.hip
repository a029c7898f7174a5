// Does a packed-fp32 instruction stream (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) always return what the scalar
// instructions return when waves of other kernels share the SIMD?  Every lane runs the same recurrence twice — once on
// float2 operands (packed instructions), once component by component (v_fma_f32 ...) — and counts the iterations after
// which the two disagree bit for bit.  tools/diag_pk.py runs it alone and under the Q-net's convolution kernels.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC -o pk_victim.so pk_victim.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
extern "C" __global__ void __launch_bounds__(128, 2) k_pk_victim(int iters, uint32_t* bad, float* sink) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f32x2 a = {1.0f + 1e-3f * (float)(t & 1023), 0.5f + 2e-3f * (float)(t & 511)};
  f32x2 b = {0.999f, 1.001f}, c = {1e-4f, -2e-4f};
  float a0 = a.x, a1 = a.y;
  uint32_t n = 0;
  for (int i = 0; i < iters; ++i) {
    // packed: v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32
    f32x2 p = __builtin_elementwise_fma(a, b, c);
    p = p * b;
    p = p + c;
    // scalar: the same operations per component (opaque to the vectoriser)
    float s0 = __builtin_fmaf(a0, b.x, c.x), s1 = __builtin_fmaf(a1, b.y, c.y);
    asm volatile("" : "+v"(s0)); asm volatile("" : "+v"(s1));
    s0 = s0 * b.x; s1 = s1 * b.y;
    asm volatile("" : "+v"(s0)); asm volatile("" : "+v"(s1));
    s0 = s0 + c.x; s1 = s1 + c.y;
    asm volatile("" : "+v"(s0)); asm volatile("" : "+v"(s1));
    n += (__float_as_uint(p.x) != __float_as_uint(s0)) | (__float_as_uint(p.y) != __float_as_uint(s1));
    a = p; a0 = s0; a1 = s1;
    if ((i & 255) == 255) { a.x = a0 = 1.0f + 1e-3f * (float)((t + i) & 1023); a.y = a1 = 0.5f + 2e-3f * (float)((t ^ i) & 511); }
  }
  if (n) atomicAdd(bad, n);
  if (a.x == 12345.0f) sink[0] = a.x + a0;
}
// Variant: the high halves of the packed operands hold "don't care" bit patterns (what the SLP vectoriser leaves in the dead
// lane of a half-used packed instruction: denormals, NaNs, infinities, lane masks), only the low half is compared.
extern "C" __global__ void __launch_bounds__(128, 2) k_pk_victim_dead(int iters, uint32_t* bad, float* sink, uint32_t junk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a0 = 1.0f + 1e-3f * (float)(t & 1023);
  f32x2 a = {a0, __uint_as_float(junk ^ (uint32_t)t)};
  f32x2 b = {0.999f, __uint_as_float(junk * 3u + 1u)}, c = {1e-4f, __uint_as_float(~junk)};
  uint32_t n = 0;
  for (int i = 0; i < iters; ++i) {
    f32x2 p = __builtin_elementwise_fma(a, b, c);
    p = p * b;
    p = p + c;
    float s0 = __builtin_fmaf(a0, b.x, c.x);
    asm volatile("" : "+v"(s0));
    s0 = s0 * b.x;
    asm volatile("" : "+v"(s0));
    s0 = s0 + c.x;
    asm volatile("" : "+v"(s0));
    n += __float_as_uint(p.x) != __float_as_uint(s0);
    a = p; a0 = s0;
    if ((i & 255) == 255) { a.x = a0 = 1.0f + 1e-3f * (float)((t + i) & 1023); a.y = __uint_as_float(junk ^ (uint32_t)(t + i)); }
  }
  if (n) atomicAdd(bad, n);
  if (a.x == 12345.0f) sink[0] = a.x + a0 + a.y;
}
extern "C" int pk_victim_dead(int iters, int blocks, void* stream, void* bad, void* sink, uint32_t junk) {
  hipLaunchKernelGGL(k_pk_victim_dead, dim3(blocks), dim3(128), 0, (hipStream_t)stream, iters, (uint32_t*)bad, (float*)sink, junk);
  return (int)hipGetLastError();
}
extern "C" int pk_victim(int iters, int blocks, void* stream, void* bad, void* sink) {
  hipLaunchKernelGGL(k_pk_victim, dim3(blocks), dim3(128), 0, (hipStream_t)stream, iters, (uint32_t*)bad, (float*)sink);
  return (int)hipGetLastError();
}
