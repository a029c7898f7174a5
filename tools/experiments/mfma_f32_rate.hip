// Issue rate of v_mfma_f32_32x32x2_f32: cycles per instruction for 1..4 waves per SIMD, independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0; for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (3.0f * iters);
}
int main() {
  float* d; hipMalloc(&d, 1 << 22);
  for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
    int threads = 64 * 4 * waves_per_simd;   // one workgroup per CU
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 2000);
    hipDeviceSynchronize();
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfmas_per_simd = 3.0 * 20000 * waves_per_simd;
    printf("waves/SIMD %d: clock64 ticks per MFMA (wave 0) %.1f; wall: %.1f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz\n", waves_per_simd, h, ms * 1e6 / mfmas_per_simd, ms * 1e6 / mfmas_per_simd * 2.4);
  }
  return 0;
}
