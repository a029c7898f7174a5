// The instruction sequence of the first failing micro-test, rebuilt by hand: packed producers with an
// SGPR-pair source, consumed a few instructions later by packed instructions that read the pair half-swapped (op_sel) with a
// negated source, against the same values taken through plain moves.  Variant v inserts `s_nop v` between producer and
// consumer (v = 0: none, back to back as the compiler scheduled them).
// hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -shared -fPIC -o pk_seq.so pk_seq.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
template <int V>
__device__ __forceinline__ uint32_t seq(float a0, float a1, float x0, float x1, float k0, float k1, float o0, float o1) {
  // v[10:11] = x * k (SGPR pair); v[30:31] = v[10:11] + o; v[32:33] = o - v[10:11];
  // packed consumers: v34 = a0 - v33, v38 = a1 - v30 (half-swapped reads, negated source); scalar consumers: v40, v41
  // returns bit 0: packed != scalar consumer; bit 1: packed consumer != the value computed outside; bit 2: scalar consumer != it
  float m0 = x0 * k0, m1 = x1 * k1;
  asm volatile("" : "+v"(m0)); asm volatile("" : "+v"(m1));
  float q0 = m0 + o0, w1 = o1 - m1;
  asm volatile("" : "+v"(q0)); asm volatile("" : "+v"(w1));
  float e0 = a0 - w1, e1 = a1 - q0;
  asm volatile("" : "+v"(e0)); asm volatile("" : "+v"(e1));
  uint32_t ps, pe, se;
  asm volatile(
    "v_mov_b32 v0, %3\n v_mov_b32 v1, %4\n v_mov_b32 v10, %5\n v_mov_b32 v11, %6\n"
    "s_mov_b32 s28, %7\n s_mov_b32 s29, %8\n s_mov_b32 s30, %9\n s_mov_b32 s31, %10\n"
    "s_nop 4\n"
    "v_pk_mul_f32 v[10:11], v[10:11], s[28:29]\n"
    "v_pk_add_f32 v[30:31], v[10:11], s[30:31]\n"
    "v_pk_add_f32 v[32:33], v[10:11], s[30:31] neg_lo:[1,0] neg_hi:[1,0]\n"
    ".if %11 > 0\n s_nop %11 - 1\n .endif\n"
    "v_pk_add_f32 v[34:35], v[0:1], v[32:33] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
    "v_pk_add_f32 v[38:39], v[0:1], v[30:31] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
    "s_nop 7\n"
    "v_sub_f32 v40, v0, v33\n"
    "v_sub_f32 v41, v1, v30\n"
    "s_nop 7\n"
    "v_cmp_ne_u32 vcc, v40, v34\n v_cndmask_b32 %0, 0, 1, vcc\n v_cmp_ne_u32 vcc, v41, v38\n v_cndmask_b32 v42, 0, 1, vcc\n v_or_b32 %0, %0, v42\n"
    "v_cmp_ne_u32 vcc, v34, %12\n v_cndmask_b32 %1, 0, 1, vcc\n v_cmp_ne_u32 vcc, v38, %13\n v_cndmask_b32 v42, 0, 1, vcc\n v_or_b32 %1, %1, v42\n"
    "v_cmp_ne_u32 vcc, v40, %12\n v_cndmask_b32 %2, 0, 1, vcc\n v_cmp_ne_u32 vcc, v41, %13\n v_cndmask_b32 v42, 0, 1, vcc\n v_or_b32 %2, %2, v42\n"
    : "=&v"(ps), "=&v"(pe), "=&v"(se)
    : "v"(a0), "v"(a1), "v"(x0), "v"(x1), "s"(k0), "s"(k1), "s"(o0), "s"(o1), "n"(V), "v"(e0), "v"(e1)
    : "v0", "v1", "v10", "v11", "v30", "v31", "v32", "v33", "v34", "v35", "v38", "v39", "v40", "v41", "v42", "s28", "s29", "s30", "s31", "vcc");
  return ps | (pe << 8) | (se << 16);
}
extern "C" __global__ void __launch_bounds__(128, 2) k_pk_seq(int iters, uint32_t* bad, float* sink) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t n[4] = {0, 0, 0, 0}, npe = 0, nse = 0;
  float a0 = 1.0f + 1e-3f * (float)(t & 1023), a1 = 0.5f + 2e-3f * (float)(t & 511);
  for (int i = 0; i < iters; ++i) {
    const float x0 = (float)((t + 3 * i) & 255), x1 = (float)((t + 7 * i) & 127);
    { const uint32_t r = seq<0>(a0, a1, x0, x1, 1e-3f, 3e-3f, 0.75f, 1.25f); n[0] += r & 1u; npe += (r >> 8) & 1u; nse += (r >> 16) & 1u; }
    n[1] += seq<1>(a0, a1, x0, x1, 1e-3f, 3e-3f, 0.75f, 1.25f) & 1u;
    n[2] += seq<2>(a0, a1, x0, x1, 1e-3f, 3e-3f, 0.75f, 1.25f) & 1u;
    n[3] += seq<4>(a0, a1, x0, x1, 1e-3f, 3e-3f, 0.75f, 1.25f) & 1u;
    a0 = 1.0f + 1e-3f * (float)((t + i) & 1023); a1 = 0.5f + 2e-3f * (float)((t ^ i) & 511);
  }
  for (int f = 0; f < 4; ++f) if (n[f]) atomicAdd(bad + f, n[f]);
  if (npe) atomicAdd(bad + 4, npe);
  if (nse) atomicAdd(bad + 5, nse);
  if (a0 == 12345.0f) sink[0] = a0 + a1;
}
extern "C" int pk_seq(int iters, int blocks, void* stream, void* bad, void* sink) {
  hipLaunchKernelGGL(k_pk_seq, dim3(blocks), dim3(128), 0, (hipStream_t)stream, iters, (uint32_t*)bad, (float*)sink);
  return (int)hipGetLastError();
}
