// One wave alone on a SIMD: cycles per vector instruction for the patterns the settle solver is made of (dependent / independent
// v_fma_f32, v_pk_fma_f32, v_mov_b32, v_fmac_f32 with a DPP quad broadcast, LDS write -> read hand-off).  s_memtime ticks = shader cycles.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
#define STR2(x) #x
#define STR(x) STR2(x)
#define TIMED(name, body)                                                                                  \
  __global__ void name(unsigned long long* out, float* sink) {                                            \
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-7f;                                                  \
    __shared__ float lds[1024];                                                                            \
    lds[threadIdx.x] = a; lds[threadIdx.x + 64] = b;                                                        \
    unsigned la = (unsigned)(size_t)&lds[0] + 16 * (threadIdx.x & 15);                                      \
    unsigned long long t0, t1;                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");      \
    asm volatile(".rept " STR(N) "\n" body "\n.endr"                                                        \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(la) :: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", \
                   "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "memory");           \
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");      \
    if (threadIdx.x == 0) out[0] = t1 - t0;                                                                 \
    sink[threadIdx.x] = a + b + c;                                                                          \
  }

TIMED(k_fma_dep, "v_fma_f32 %0, %0, %1, %2")
TIMED(k_fma_ind, "v_fma_f32 v10, %0, %1, %2\n v_fma_f32 v11, %0, %1, %2\n v_fma_f32 v12, %0, %1, %2\n v_fma_f32 v13, %0, %1, %2")
TIMED(k_fma_dep2, "v_fma_f32 v10, v10, %1, %2\n v_fma_f32 v11, v11, %1, %2")
TIMED(k_pk_dep, "v_pk_fma_f32 v[10:11], v[10:11], v[12:13], v[14:15]")
TIMED(k_pk_ind, "v_pk_fma_f32 v[10:11], v[20:21], v[12:13], v[14:15]\n v_pk_fma_f32 v[16:17], v[20:21], v[12:13], v[14:15]\n v_pk_fma_f32 v[18:19], v[20:21], v[12:13], v[14:15]\n v_pk_fma_f32 v[22:23], v[20:21], v[12:13], v[14:15]")
TIMED(k_pk_bcast_dep, "v_pk_fma_f32 v[10:11], v[12:13], v[10:11], v[10:11] op_sel_hi:[1,0,1]")
TIMED(k_mov_dep, "v_mov_b32 %0, %0")
TIMED(k_mov_ind, "v_mov_b32 v10, %0\n v_mov_b32 v11, %1\n v_mov_b32 v12, %2\n v_mov_b32 v13, %0")
TIMED(k_fmac_dpp_ind, "v_fmac_f32_dpp v10, %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v11, %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v12, %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp v13, %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf")
TIMED(k_fmac_dpp_dep, "v_fmac_f32_dpp %2, %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf")
TIMED(k_fmac_dep, "v_fmac_f32 %2, %0, %1")
TIMED(k_mix_row, "v_pk_mul_f32 v[10:11], v[12:13], v[14:15]\n v_pk_fma_f32 v[10:11], v[16:17], v[18:19], v[10:11]\n v_pk_fma_f32 v[10:11], v[20:21], v[22:23], v[10:11]\n v_add_f32 v24, v10, v11\n v_sub_f32 v24, v24, %0\n v_fma_f32 v24, -v24, %1, %2\n v_add_f32 v24, v25, v24\n v_med3_f32 v26, v24, 0, %1\n v_sub_f32 v24, v26, v25\n v_pk_fma_f32 v[14:15], v[28:29], v[24:25], v[14:15] op_sel_hi:[1,0,1]")
TIMED(k_lds_handoff, "ds_write_b128 %3, v[10:13]\n ds_write_b64 %3, v[14:15] offset:16\n ds_read_b128 v[10:13], %3\n ds_read_b64 v[14:15], %3 offset:16\n s_waitcnt lgkmcnt(0)\n v_add_f32 v10, v10, v14")
TIMED(k_lds_read, "ds_read_b128 v[10:13], %3\n s_waitcnt lgkmcnt(0)\n v_add_f32 v14, v10, v11")
TIMED(k_dpp_mov12, "v_mov_b32_dpp v10, v20 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v11, v21 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v12, v22 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v13, v23 row_shr:1 row_mask:0xf bank_mask:0xf")

// control flow: a taken backward branch per iteration, and the exec-mask guard of a turn
__global__ void k_loop_branch(unsigned long long* out, float* sink) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-7f;
  unsigned long long t0, t1; int n = N;
  asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  asm volatile("1:\n v_fmac_f32 %2, %0, %1\n s_add_i32 %3, %3, -1\n s_cmp_eq_u32 %3, 0\n s_cbranch_scc0 1b" : "+v"(a), "+v"(b), "+v"(c), "+s"(n) :: "scc");
  asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = a + b + c;
}
TIMED(k_guard, "v_subrev_co_u32 %3, vcc, 1, %3\n s_and_saveexec_b64 s[20:21], vcc\n v_fmac_f32 %2, %0, %1\n s_or_b64 exec, exec, s[20:21]")
TIMED(k_guard_branch, "v_subrev_co_u32 %3, vcc, 1, %3\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 2f\n v_fmac_f32 %2, %0, %1\n2:\n s_or_b64 exec, exec, s[20:21]")

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8); hipMalloc(&sink, 4096);
  struct { const char* name; void (*k)(unsigned long long*, float*); int per; } ks[] = {
    {"v_fma_f32 dependent chain", k_fma_dep, 1}, {"v_fma_f32 independent (4 dsts)", k_fma_ind, 4}, {"v_fma_f32 two interleaved chains", k_fma_dep2, 2},
    {"v_pk_fma_f32 dependent chain", k_pk_dep, 1}, {"v_pk_fma_f32 independent (4 dsts)", k_pk_ind, 4}, {"v_pk_fma_f32 dependent, op_sel broadcast", k_pk_bcast_dep, 1},
    {"v_mov_b32 dependent", k_mov_dep, 1}, {"v_mov_b32 independent", k_mov_ind, 4},
    {"v_fmac_f32_dpp quad_perm independent", k_fmac_dpp_ind, 4}, {"v_fmac_f32_dpp quad_perm dependent", k_fmac_dpp_dep, 1}, {"v_fmac_f32 dependent", k_fmac_dep, 1},
    {"one pair-row-like chain (10 instr)", k_mix_row, 10}, {"LDS hand-off: write b128+b64, read back, wait, add (6 instr)", k_lds_handoff, 6},
    {"ds_read_b128 + wait + add (3 instr)", k_lds_read, 3}, {"loop: fmac + s_add + s_cmp + taken branch (per iteration)", k_loop_branch, 1},
    {"guard: v_subrev_co + s_and_saveexec + fmac + s_or exec (per group)", k_guard, 1}, {"guard with s_cbranch_execz (mostly taken) (per group)", k_guard_branch, 1}, {"v_mov_b32_dpp row_shr:1 independent", k_dpp_mov12, 4}};
  for (auto& e : ks) {
    unsigned long long best = ~0ull;
    for (int r = 0; r < 5; ++r) {
      hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, out, sink);
      unsigned long long t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
      if (t < best) best = t;
    }
    printf("%-62s %7.2f cycles per instruction (%llu ticks / %d)\n", e.name, (double)best / (N * e.per), best, N * e.per);
  }
  return 0;
}
