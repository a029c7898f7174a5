// Diagnostic: fills the LDS of every CU (and a stretch of registers) with a pattern, so that a kernel that reads LDS it
// never wrote shows up as a result that depends on the pattern.  hipcc --offload-arch=gfx950 -shared -fPIC -o lds_poison.so
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void __launch_bounds__(256) k_lds_poison(uint32_t pattern, uint32_t* sink) {
  extern __shared__ uint32_t lds[];
  const int n = 40 * 1024 / 4;   // 40 KB per workgroup: four per CU cover the 160 KB
  for (int i = threadIdx.x; i < n; i += 256) lds[i] = pattern ^ (pattern == 0xdeadbeefu ? (uint32_t)i * 2654435761u : 0u);
  __syncthreads();
  uint32_t s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += lds[i];
  // hold the CU for a while so that the four workgroups of a CU are resident together
  for (int k = 0; k < 2000; ++k) s = s * 1664525u + 1013904223u;
  if (s == 12345u && sink) sink[0] = s;
}
extern "C" int lds_poison(uint32_t pattern, int blocks, void* stream, void* sink) {
  hipFuncSetAttribute((const void*)k_lds_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  hipLaunchKernelGGL(k_lds_poison, dim3(blocks), dim3(256), 40 * 1024, (hipStream_t)stream, pattern, (uint32_t*)sink);
  return (int)hipGetLastError();
}
