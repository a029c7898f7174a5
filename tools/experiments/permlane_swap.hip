// What v_permlane16_swap_b32 (gfx950) returns when both operands hold the same value: lane -> (r[0], r[1]).
// build: hipcc --offload-arch=gfx950 -O2 tools/experiments/permlane_swap.hip -o gpurun_out/permlane_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned x = threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  out[2 * threadIdx.x] = r[0]; out[2 * threadIdx.x + 1] = r[1];
  // the DPP controls used for 16-lane reductions
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  out[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false);  // row_half_mirror
  out[320 + threadIdx.x] = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false);  // row_mirror
}
int main() {
  unsigned* d; unsigned h[384];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("swap r0:"); for (int i = 0; i < 64; ++i) printf(" %u", h[2 * i]); printf("\nswap r1:"); for (int i = 0; i < 64; ++i) printf(" %u", h[2 * i + 1]);
  const char* nm[4] = {"qp1032", "qp2301", "halfmir", "rowmir"};
  for (int t = 0; t < 4; ++t) { printf("\n%s:", nm[t]); for (int i = 0; i < 64; ++i) printf(" %u", h[128 + 64 * t + i]); }
  printf("\n");
  return 0;
}
