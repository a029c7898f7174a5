// The ray-cast inner loop in isolation: per iteration 5 LDS reads (A value, 16 C values), 3 v_mfma_f32_32x32x2_f32 sharing
// one C and writing three D sets, 24 v_min3_f32.  ns per MFMA per SIMD at 4 waves per SIMD, for a few variants.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int VAR>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  __shared__ float pc[4096], pa[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) { pc[i] = i * 1e-3f; pa[i] = i * 1e-4f; }
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  float m0 = 1e30f, m1 = 1e30f, m2 = 1e30f;
  const float b0 = lane * 0.1f, b1 = lane * 0.2f, b2 = lane * 0.3f;
  for (int i = 0; i < iters; ++i) {
    const int sb = (i * 32) & 4095 & ~31;
    const float av = pa[sb + j];
    f32x16 cv;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float4 c4 = *(const float4*)(pc + sb + 8 * q + 4 * h); cv[4*q] = c4.x; cv[4*q+1] = c4.y; cv[4*q+2] = c4.z; cv[4*q+3] = c4.w; }
    f32x16 d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, cv, 0, 0, 0);
    f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, cv, 0, 0, 0);
    f32x16 d2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b2, cv, 0, 0, 0);
    if (VAR == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { m0 = __builtin_amdgcn_fmed3f(m0, d0[r], -__builtin_inff()); m1 = __builtin_amdgcn_fmed3f(m1, d1[r], -__builtin_inff()); m2 = __builtin_amdgcn_fmed3f(m2, d2[r], -__builtin_inff()); }
    } else {   // VAR 1: no reductions (only one element of each D is kept alive)
      m0 = __builtin_amdgcn_fmed3f(m0, d0[0], -__builtin_inff()); m1 = __builtin_amdgcn_fmed3f(m1, d1[5], -__builtin_inff()); m2 = __builtin_amdgcn_fmed3f(m2, d2[9], -__builtin_inff());
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = m0 + m1 + m2;
}
template <int VAR> void run(const char* tag, float* d) {
  hipLaunchKernelGGL(k<VAR>, dim3(512), dim3(512), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 5000;
  hipEventRecord(e0); hipLaunchKernelGGL(k<VAR>, dim3(512), dim3(512), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // 512 workgroups of 8 waves on 256 CUs: 2 per CU = 4 waves per SIMD; MFMAs per SIMD = 4 waves x 3 x iters
  printf("%s: %.1f ns per MFMA per SIMD\n", tag, ms * 1e6 / (4.0 * 3 * iters));
}
int main() {
  float* d; hipMalloc(&d, 512 * 512 * 4);
  run<0>("3 MFMA + 24 min3 + LDS operands", d);
  run<1>("3 MFMA + LDS operands, no reductions", d);
  return 0;
}
