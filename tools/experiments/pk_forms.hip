// Which FORM of a packed-fp32 instruction returns a wrong result beside an MFMA kernel?  Each
// form the SLP vectoriser emitted in the env kernels — source modifiers (neg_lo / neg_hi), half selection (op_sel /
// op_sel_hi), an SGPR pair or an inline constant as a source — is evaluated by inline assembly on fixed registers and
// compared, bit for bit, with the scalar instructions that define it.  bad[f] counts the mismatches of form f.
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o pk_forms.so pk_forms.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#define CHECK(f, r0, r1, e0, e1) n[f] += (__float_as_uint(r0) != __float_as_uint(e0)) | (__float_as_uint(r1) != __float_as_uint(e1))
extern "C" __global__ void __launch_bounds__(128, 2) k_pk_forms(int iters, uint32_t* bad, float* sink) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a0 = 1.0f + 1e-3f * (float)(t & 1023), a1 = 0.5f + 2e-3f * (float)(t & 511);
  float b0 = 0.75f - 1e-3f * (float)(t & 255), b1 = 1.25f + 3e-3f * (float)(t & 127);
  float c0 = 0.125f, c1 = -0.375f;
  uint32_t n[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    float r0, r1;
    // A: a - b on both halves through neg modifiers
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v12, %4\n v_mov_b32 v13, %5\n"
                 "v_pk_add_f32 v[14:15], v[10:11], v[12:13] neg_lo:[0,1] neg_hi:[0,1]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15");
    { float e0 = a0 - b0, e1 = a1 - b1; CHECK(0, r0, r1, e0, e1);
      if ((__float_as_uint(r0) != __float_as_uint(e0) || __float_as_uint(r1) != __float_as_uint(e1)) && atomicAdd(bad + 15, 1u) == 0u) {
        sink[8] = a0; sink[9] = a1; sink[10] = b0; sink[11] = b1; sink[12] = r0; sink[13] = r1; sink[14] = e0; sink[15] = e1; } }
    // G: neg on the FIRST source (VGPR): b - a
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v12, %4\n v_mov_b32 v13, %5\n"
                 "v_pk_add_f32 v[14:15], v[10:11], v[12:13] neg_lo:[1,0] neg_hi:[1,0]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15");
    { float e0 = b0 - a0, e1 = b1 - a1; CHECK(6, r0, r1, e0, e1); }
    // H: packed multiply with a negated second source
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v12, %4\n v_mov_b32 v13, %5\n"
                 "v_pk_mul_f32 v[14:15], v[10:11], v[12:13] neg_lo:[0,1] neg_hi:[0,1]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15");
    { float e0 = a0 * -b0, e1 = a1 * -b1; CHECK(7, r0, r1, e0, e1); }
    // I: packed fma with a negated addend: a b - c
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v12, %4\n v_mov_b32 v13, %5\n v_mov_b32 v16, %6\n v_mov_b32 v17, %7\n"
                 "v_pk_fma_f32 v[14:15], v[10:11], v[12:13], v[16:17] neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    { float e0 = __builtin_fmaf(a0, b0, -c0), e1 = __builtin_fmaf(a1, b1, -c1); CHECK(8, r0, r1, e0, e1); }
    // J: plain packed add, no modifier (control)
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v12, %4\n v_mov_b32 v13, %5\n"
                 "v_pk_add_f32 v[14:15], v[10:11], v[12:13]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15");
    { float e0 = a0 + b0, e1 = a1 + b1; CHECK(9, r0, r1, e0, e1); }
    // N0: a packed producer, 0 independent instructions, then a packed consumer that reads the pair half-swapped
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v16, %4\n v_mov_b32 v17, %5\n v_mov_b32 v18, %6\n v_mov_b32 v19, %7\n"
                 "v_pk_add_f32 v[12:13], v[16:17], v[18:19]\n"
                 "v_pk_add_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22");
    { float p0 = b0 + c0, p1 = b1 + c1; asm volatile("" : "+v"(p0)); asm volatile("" : "+v"(p1));
      float e0 = a0 - p1, e1 = a1 - p0; CHECK(13, r0, r1, e0, e1); }
    // N1: a packed producer, 1 independent instructions, then a packed consumer that reads the pair half-swapped
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v16, %4\n v_mov_b32 v17, %5\n v_mov_b32 v18, %6\n v_mov_b32 v19, %7\n"
                 "v_pk_add_f32 v[12:13], v[16:17], v[18:19]\n v_mov_b32 v20, v10\n"
                 "v_pk_add_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22");
    { float p0 = b0 + c0, p1 = b1 + c1; asm volatile("" : "+v"(p0)); asm volatile("" : "+v"(p1));
      float e0 = a0 - p1, e1 = a1 - p0; CHECK(14, r0, r1, e0, e1); }
    // N2: a packed producer, 2 independent instructions, then a packed consumer that reads the pair half-swapped
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v16, %4\n v_mov_b32 v17, %5\n v_mov_b32 v18, %6\n v_mov_b32 v19, %7\n"
                 "v_pk_add_f32 v[12:13], v[16:17], v[18:19]\n v_mov_b32 v20, v10\n v_mov_b32 v21, v10\n"
                 "v_pk_add_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22");
    { float p0 = b0 + c0, p1 = b1 + c1; asm volatile("" : "+v"(p0)); asm volatile("" : "+v"(p1));
      float e0 = a0 - p1, e1 = a1 - p0; CHECK(15, r0, r1, e0, e1); }
    // N3: a packed producer, 3 independent instructions, then a packed consumer that reads the pair half-swapped
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v16, %4\n v_mov_b32 v17, %5\n v_mov_b32 v18, %6\n v_mov_b32 v19, %7\n"
                 "v_pk_add_f32 v[12:13], v[16:17], v[18:19]\n v_mov_b32 v20, v10\n v_mov_b32 v21, v10\n v_mov_b32 v22, v10\n"
                 "v_pk_add_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22");
    { float p0 = b0 + c0, p1 = b1 + c1; asm volatile("" : "+v"(p0)); asm volatile("" : "+v"(p1));
      float e0 = a0 - p1, e1 = a1 - p0; CHECK(16, r0, r1, e0, e1); }
    // K: the scalar subtraction as the compiler's own instruction against the VOP3 form with a neg modifier (control)
    asm volatile("v_add_f32_e64 %0, %1, -%2\n" : "=v"(r0) : "v"(a0), "v"(b0));
    { float e0 = a0 - b0; asm volatile("" : "+v"(e0)); CHECK(10, r0, r0, e0, e0); }
    // B: (a0 b0, a1 b0): the low half of the second source for both
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v12, %4\n v_mov_b32 v13, %5\n"
                 "v_pk_mul_f32 v[14:15], v[10:11], v[12:13] op_sel_hi:[1,0]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15");
    { float e0 = a0 * b0, e1 = a1 * b0; CHECK(1, r0, r1, e0, e1); }
    // C: + 0.5 as an inline constant
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n"
                 "v_pk_add_f32 v[14:15], v[10:11], 0.5 op_sel_hi:[1,0]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1) : "v10", "v11", "v14", "v15");
    { float e0 = a0 + 0.5f, e1 = a1 + 0.5f; CHECK(2, r0, r1, e0, e1); }
    // D: fma with the high half of the first source for the low lane: (a1 b0 + c0, a1 b1 + c0)
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v12, %4\n v_mov_b32 v13, %5\n v_mov_b32 v16, %6\n v_mov_b32 v17, %7\n"
                 "v_pk_fma_f32 v[14:15], v[10:11], v[12:13], v[16:17] op_sel:[1,0,0] op_sel_hi:[1,1,0]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    { float e0 = __builtin_fmaf(a1, b0, c0), e1 = __builtin_fmaf(a1, b1, c0); CHECK(3, r0, r1, e0, e1); }
    // E: SGPR pair source, first source negated: (s - a)
    {
      const float s0 = 1.0f, s1 = 2.0f;
      asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n s_mov_b32 s20, %4\n s_mov_b32 s21, %5\n"
                   "v_pk_add_f32 v[14:15], v[10:11], s[20:21] neg_lo:[1,0] neg_hi:[1,0]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "s"(s0), "s"(s1) : "v10", "v11", "v14", "v15", "s20", "s21");
      float e0 = s0 - a0, e1 = s1 - a1; CHECK(4, r0, r1, e0, e1);
    }
    // F: -x - 0: both sources negated, zero as an inline constant (how the vectoriser negates a pair)
    asm volatile("v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n"
                 "v_pk_add_f32 v[14:15], v[10:11], 0 neg_lo:[1,1] neg_hi:[1,1]\n v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n"
                 : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1) : "v10", "v11", "v14", "v15");
    { float e0 = -a0 - 0.0f, e1 = -a1 - 0.0f; CHECK(5, r0, r1, e0, e1); }
    a0 = 1.0f + 1e-3f * (float)((t + i) & 1023); a1 = 0.5f + 2e-3f * (float)((t ^ i) & 511);
    b0 = 0.75f - 1e-3f * (float)((t + 3 * i) & 255); b1 = 1.25f + 3e-3f * (float)((t + 7 * i) & 127);
  }
  for (int f = 0; f < 12; ++f) if (n[f]) atomicAdd(bad + f, n[f]);
  if (a0 == 12345.0f) sink[0] = a0 + a1;
}
extern "C" int pk_forms(int iters, int blocks, void* stream, void* bad, void* sink) {
  hipLaunchKernelGGL(k_pk_forms, dim3(blocks), dim3(128), 0, (hipStream_t)stream, iters, (uint32_t*)bad, (float*)sink);
  return (int)hipGetLastError();
}
