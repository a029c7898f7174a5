// Does the hardware stop a workgroup from touching LDS beyond its own allocation?  Victim workgroups fill their LDS with a
// pattern, wait, and check it; attacker workgroups with a small allocation write beyond it — whole dwords far outside
// (mode 0) or 8- / 16-byte stores that START inside the allocation and END outside it (mode 1 / 2).
// hipcc --offload-arch=gfx950 -O2 -o lds_oob lds_oob.hip && ./lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void __launch_bounds__(64) victim(int words, uint32_t* bad, long long spin) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < words; i += 64) lds[i] = 0xA5A50000u + i;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(32);
  __syncthreads();
  uint32_t n = 0;
  for (int i = threadIdx.x; i < words; i += 64) n += lds[i] != 0xA5A50000u + i;
  if (n) atomicAdd(bad, n);
}
__global__ void __launch_bounds__(64) attacker(int mode, int alloc_words, int reach_words, uint32_t* sink) {
  extern __shared__ uint32_t lds[];
  uint32_t s = 0;
  if (mode == 0) {
    volatile uint32_t* p = lds;
    for (int i = threadIdx.x; i < reach_words; i += 64) p[i] = 0xDEAD0000u + i;
    for (int i = threadIdx.x; i < reach_words; i += 64) s += p[i];
  } else if (mode == 1) {      // 8-byte store whose second dword lies outside (allocation = odd number of dwords)
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(lds + alloc_words - 1);
    const uint64_t v = 0xDEAD0002DEAD0001ull;
    if (threadIdx.x == 0) asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory");
  } else {                     // 16-byte store starting 1 dword before the end
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(lds + alloc_words - 1);
    const u32x4 v = {0xDEAD0001u, 0xDEAD0002u, 0xDEAD0003u, 0xDEAD0004u};
    if (threadIdx.x == 0) asm volatile("ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory");
  }
  if (s == 1u) sink[0] = s;
}
int main() {
  uint32_t *bad, *sink; (void)hipMalloc(&bad, 4); (void)hipMalloc(&sink, 4);
  hipStream_t a, b; (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  (void)hipFuncSetAttribute((const void*)victim, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int mode = 0; mode < 3; ++mode) {
    (void)hipMemset(bad, 0, 4);
    (void)hipDeviceSynchronize();
    // victims of 10 KB (16 per CU) so that attackers' allocations sit between them; 20 ms at the 100 MHz wall clock
    hipLaunchKernelGGL(victim, dim3(3000), dim3(64), 10 * 1024, a, 2560, bad, 2000000LL);
    const int alloc_words = mode == 0 ? 256 : 255;      // 1020 bytes: not a multiple of 8
    for (int k = 0; k < 300; ++k) hipLaunchKernelGGL(attacker, dim3(4096), dim3(64), 4 * alloc_words, b, mode, alloc_words, 40 * 1024, sink);
    (void)hipDeviceSynchronize();
    uint32_t h = 0; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("mode %d: victim words corrupted by other workgroups' out-of-allocation LDS stores: %u (%s)\n", mode, h, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
