// srl_bf16.h — float32 -> bfloat16 conversions of the Q-net kernels (libstackrl_qnet.so), gfx950.
// Round to nearest even with v_cvt_pk_bf16_f32 (two values per instruction); until round 4 these were integer sequences
// ((u + 0x7fff + ((u >> 16) & 1)) >> 16: three instructions per value).  Same results for finite values.
// The fp32-class ("bf16x3") kernels split an operand as x = hi + lo + O(2^-17 |x|), hi = bf16(x), lo = bf16(x - hi).
#ifndef SRL_BF16_H_
#define SRL_BF16_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 srl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float srl_f32x2 __attribute__((ext_vector_type(2)));

// bf16(a) in the low half, bf16(b) in the high half
__device__ __forceinline__ uint32_t srl_pk_bf16(float a, float b) {
  const srl_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, srl_bf16x2));
}

// bf16(a), zero-extended
__device__ __forceinline__ uint32_t srl_bf16(float a) { return srl_pk_bf16(a, 0.0f); }

// hi = {bf16(a), bf16(b)}, lo = {bf16(a - hi_a), bf16(b - hi_b)}
__device__ __forceinline__ void srl_split_bf16(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = srl_pk_bf16(a, b);
  lo = srl_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

#endif
