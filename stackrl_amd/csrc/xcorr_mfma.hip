// xcorr_mfma.hip — K6 on the matrix cores: per-sample VALID cross-correlation of the Q-net (layers.py:21-38), its
// two gradients, gfx950.
//
//   forward  out[b, y, x]   = sum_{c, i, t} X[b, c, y + i, x + t] * W[b, c, i, t]        X [C, H, H], W [C, kh, kh]
//   d/dX     dX[b, c, y, x] = sum_{i, t} P[b, y + i, x + t] * Wf[b, c, i, t]             P = dOut zero-padded by kh - 1,
//                                                                                         Wf = W flipped in both axes
//   d/dW     dW[b, c, i, t] = sum_{y, x} X[b, c, i + y, t + x] * dOut[b, y, x]
//
// All three are "correlate a map with a per-sample kernel", differing in which operand is per channel and whether the
// channels are summed, so one kernel family serves them.  The op has no batch-shared operand (every sample brings its
// own kernel), so it is not a library GEMM; the library route (a grouped convolution with B groups) costs 6 ms per
// forward at B = 32.  It is made MFMA-shaped per kernel row i: the 1-D correlation along x is a product with a banded
// Toeplitz matrix,
//
//   out[y, x] += sum_k A[y, k] * T[k, x],     A[y, k] = map[y + i, x0 + k],     T[k, x] = kern[i, k - x]  (0 <= k - x < KH)
//
// For a 16 x 16 output tile at (y0, x0) only k in [0, 16 + KH - 1) matters: KB = 1 + ceil((KH - 1) / 16) k-blocks of 16
// (v_mfma_f32_16x16x16_bf16, fp32 accumulation).  T does not depend on the tile (shift invariance), so its MFMA
// fragments are built once per kernel row by a small prep kernel (k_toeplitz) and then feed every tile.
//
// Main kernel: one workgroup per sample, one wave per row of output tiles.  One channel of the map is staged in LDS at
// a time (bf16, zero-padded); per kernel row a wave reads its A fragments (8 bytes per lane, shared by its tiles: tile
// t uses k-blocks t .. t + KB - 1), the KB Toeplitz fragments (global, L2-resident: all waves of the sample read the
// same ones) and issues T * KB MFMAs.  Accumulators stay in registers over all kernel rows (and channels, when summed).
//
// Precision: 0 = operands rounded to bf16 (products exact in fp32, fp32 accumulation) — the rollout path under bf16
// autocast; 1 = "bf16x3": every fp32 operand is split into hi + lo bf16 parts and hi*hi + hi*lo + lo*hi is
// accumulated (the dropped terms are below 2^-16 relative per product), which is fp32-class accuracy at a third of the
// bf16 MFMA rate and still an order of magnitude above the fp32 vector rate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stackrl_qnet.h"

namespace {

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int HIN, int KH>
struct XcorrCfg {
  static constexpr int O = HIN - KH + 1;             // outputs per side
  static constexpr int T = (O + 15) / 16;            // 16 x 16 output tiles per side = waves per workgroup
  static constexpr int KB = 1 + (KH - 1 + 15) / 16;  // k-blocks per tile and kernel row
  static constexpr int NKB = T + KB - 1;             // k-blocks a row of tiles touches
  static constexpr int ROWS = 16 * T + KH - 1;       // staged rows; rows >= HIN stay zero
  static constexpr int RS = 16 * NKB + 4;            // row stride in elements: rows stay 8-byte aligned
};

__device__ __forceinline__ uint32_t bf16_rne(float f) {   // round to nearest even (finite inputs)
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

template <bool F32>
__device__ __forceinline__ float load_elem(const void* p, size_t k) {
  if (F32) return ((const float*)p)[k];
  return bf16_to_f32(((const uint16_t*)p)[k]);
}

// Toeplitz fragments of kernel row `row` (B operand of v_mfma_f32_16x16x16_bf16: lane l holds column l % 16, rows
// 4 (l / 16) .. + 3 of the k-block): frag[row][j][lane][r] = kern[row][t], t = 16 j + 4 (lane / 16) + r - lane % 16,
// zero outside [0, KH).  SPLIT: a second array of the same size holds the lo parts.
template <int KH, bool F32, bool SPLIT>
__global__ void __launch_bounds__(64) k_toeplitz(const void* __restrict__ kern, uint16_t* __restrict__ frag, size_t nrows) {
  constexpr int KB = 1 + (KH - 1 + 15) / 16;
  const size_t row = blockIdx.x;
  const int j = blockIdx.y, lane = threadIdx.x;
  uint32_t hi[4], lo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = 16 * j + 4 * (lane >> 4) + r - (lane & 15);
    const float v = (t >= 0 && t < KH) ? load_elem<F32>(kern, row * KH + t) : 0.0f;
    hi[r] = bf16_rne(v);
    lo[r] = SPLIT ? bf16_rne(v - bf16_to_f32(hi[r])) : 0u;
  }
  const size_t at = (row * KB + j) * 64 + lane;
  ((uint2*)frag)[at] = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
  if (SPLIT) ((uint2*)frag)[nrows * KB * 64 + at] = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
}

// IN_PER_C: the map is per channel ([B][C][HIN][HIN]) or shared by the channels ([B][HIN][HIN]);
// K_PER_C: likewise for the kernel rows; SUM: one output per sample (channels summed) or one per (sample, channel).
template <int HIN, int KH, bool IN_PER_C, bool K_PER_C, bool SUM, bool F32, bool SPLIT>
__global__ void __launch_bounds__((64 * XcorrCfg<HIN, KH>::T))
k_xcorr_mfma(const void* __restrict__ in, const uint16_t* __restrict__ frag, float* __restrict__ out, int C, int cper,
             size_t nrows) {
  typedef XcorrCfg<HIN, KH> G;
  extern __shared__ uint16_t xs[];   // [ROWS][RS] bf16 hi parts (+ the same again for the lo parts)
  constexpr int TILE = G::ROWS * G::RS;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, yt = tid >> 6;
  constexpr int NT = 64 * G::T;
  {   // zero the tile(s) once: the padding rows / columns are never written again
    uint2* z = (uint2*)xs;
    for (int k = tid; k < (SPLIT ? 2 : 1) * TILE / 4; k += NT) z[k] = make_uint2(0u, 0u);
  }
  f32x4 acc[G::T];
#pragma unroll
  for (int t = 0; t < G::T; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const int arow = 16 * yt + (lane & 15), acol = 4 * (lane >> 4);
  const size_t lo_off = nrows * G::KB * 64;   // in fragments (8 bytes)
  // blockIdx.y takes channels [c0, c1): with few samples the channels are spread over workgroups (summed outputs
  // then land in per-workgroup partials, reduced by k_sum_partials in a fixed order)
  const int c0 = blockIdx.y * cper, c1 = min(C, c0 + cper);
  for (int c = c0; c < c1; ++c) {
    if (IN_PER_C || c == c0) {
      __syncthreads();   // the previous channel's reads are done (first trip: the zero fill is complete)
      const size_t base = (IN_PER_C ? (size_t)b * C + c : (size_t)b) * HIN * HIN;
      if (HIN % 4 == 0) {   // 4 elements per thread and step
        for (int k = tid; k < HIN * HIN / 4; k += NT) {
          const int r = k / (HIN / 4), cc = (k - r * (HIN / 4)) * 4;
          float v[4];
          if (F32) {
            const float4 q = ((const float4*)((const float*)in + base))[k];
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
          } else {
            const uint2 q = ((const uint2*)((const uint16_t*)in + base))[k];
            v[0] = bf16_to_f32(q.x & 0xffffu); v[1] = bf16_to_f32(q.x >> 16);
            v[2] = bf16_to_f32(q.y & 0xffffu); v[3] = bf16_to_f32(q.y >> 16);
          }
          uint32_t hi[4], lo[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hi[e] = bf16_rne(v[e]);
            lo[e] = SPLIT ? bf16_rne(v[e] - bf16_to_f32(hi[e])) : 0u;
          }
          *(uint2*)(xs + r * G::RS + cc) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
          if (SPLIT) *(uint2*)(xs + TILE + r * G::RS + cc) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
        }
      } else {   // odd sides (the padded gradient map): element by element
        for (int k = tid; k < HIN * HIN; k += NT) {
          const int r = k / HIN, cc = k - r * HIN;
          const float v = load_elem<F32>(in, base + k);
          const uint32_t hi = bf16_rne(v);
          xs[r * G::RS + cc] = (uint16_t)hi;
          if (SPLIT) xs[TILE + r * G::RS + cc] = (uint16_t)bf16_rne(v - bf16_to_f32(hi));
        }
      }
      __syncthreads();
    }
    const bf16x4* tw = (const bf16x4*)frag + ((K_PER_C ? (size_t)b * C + c : (size_t)b) * KH) * G::KB * 64 + lane;
    // kernel-row loop, software-pipelined: the Toeplitz fragments of row i + 1 are requested (global, L2) before the
    // MFMAs of row i are issued, so their latency is covered by the 21 MFMAs instead of being waited for
    bf16x4 tf[G::KB], tl[SPLIT ? G::KB : 1];
#pragma unroll
    for (int j = 0; j < G::KB; ++j) {
      tf[j] = tw[(size_t)j * 64];
      if (SPLIT) tl[j] = tw[lo_off + (size_t)j * 64];
    }
#pragma unroll 2
    for (int i = 0; i < KH; ++i) {
      const int in = i + 1 < KH ? i + 1 : i;
      bf16x4 nf[G::KB], nl[SPLIT ? G::KB : 1];
#pragma unroll
      for (int j = 0; j < G::KB; ++j) {
        nf[j] = tw[((size_t)in * G::KB + j) * 64];
        if (SPLIT) nl[j] = tw[lo_off + ((size_t)in * G::KB + j) * 64];
      }
      bf16x4 af[G::NKB], al[SPLIT ? G::NKB : 1];
      const uint16_t* ar = xs + (arow + i) * G::RS + acol;
#pragma unroll
      for (int kb = 0; kb < G::NKB; ++kb) {
        af[kb] = *(const bf16x4*)(ar + 16 * kb);
        if (SPLIT) al[kb] = *(const bf16x4*)(ar + TILE + 16 * kb);
      }
#pragma unroll
      for (int t = 0; t < G::T; ++t)
#pragma unroll
        for (int j = 0; j < G::KB; ++j) {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af[t + j], tf[j], acc[t], 0, 0, 0);
          if (SPLIT) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af[t + j], tl[j], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[t + j], tf[j], acc[t], 0, 0, 0);
          }
        }
#pragma unroll
      for (int j = 0; j < G::KB; ++j) {
        tf[j] = nf[j];
        if (SPLIT) tl[j] = nl[j];
      }
    }
    if (!SUM || c == c1 - 1) {
      // D fragment: lane l holds column l % 16, rows 4 (l / 16) .. + 3 of the tile
      float* ob = out + (SUM ? (size_t)b * gridDim.y + blockIdx.y : (size_t)b * C + c) * G::O * G::O;
#pragma unroll
      for (int t = 0; t < G::T; ++t) {
        const int xcol = 16 * t + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int y = 16 * yt + 4 * (lane >> 4) + r;
          if (y < G::O && xcol < G::O) ob[(size_t)y * G::O + xcol] = acc[t][r];
        }
        acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
  }
}

// out[b][k] = partial[b][0][k] + partial[b][1][k] + ... (fixed order)
__global__ void __launch_bounds__(256) k_sum_partials(const float* __restrict__ partial, float* __restrict__ out, int P, int n) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const float* p = partial + (size_t)b * P * n + k;
  float acc = p[0];
  for (int q = 1; q < P; ++q) acc += p[(size_t)q * n];
  out[(size_t)b * n + k] = acc;
}

// channel split: enough workgroups to cover the 256 CUs when the batch is small
inline void channel_split(int B, int C, int* cper, int* csplit) {
  int want = (256 + B - 1) / B;
  if (want > C) want = C;
  if (want < 1) want = 1;
  *cper = (C + want - 1) / want;
  *csplit = (C + *cper - 1) / *cper;
}

thread_local char x_err[256] = "";

template <int HIN, int KH, bool IN_PER_C, bool K_PER_C, bool SUM, bool IF32, bool KF32, bool SPLIT>
int launch2(const void* in, const void* kern, float* out, void* scratch, int B, int C, hipStream_t st) {
  typedef XcorrCfg<HIN, KH> G;
  const size_t nrows = (size_t)(K_PER_C ? B * C : B) * KH;
  hipLaunchKernelGGL((k_toeplitz<KH, KF32, SPLIT>), dim3((unsigned)nrows, G::KB), dim3(64), 0, st, kern, (uint16_t*)scratch, nrows);
  const size_t lds = sizeof(uint16_t) * G::ROWS * G::RS * (SPLIT ? 2 : 1);
  auto fn = k_xcorr_mfma<HIN, KH, IN_PER_C, K_PER_C, SUM, IF32, SPLIT>;
  static bool lds_opted_in = false;   // per instantiation; set once, outside any later stream capture
  if (lds > 65536 && !lds_opted_in) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: LDS opt-in of %zu bytes failed: %s", lds, hipGetErrorString(e));
      return 2;
    }
    lds_opted_in = true;
  }
  int cper, csplit;
  channel_split(B, C, &cper, &csplit);
  const size_t frag_bytes = nrows * G::KB * 64 * 4 * sizeof(uint16_t) * (SPLIT ? 2 : 1);
  float* partial = (float*)((char*)scratch + frag_bytes);
  const bool two_pass = SUM && csplit > 1;
  hipLaunchKernelGGL(fn, dim3(B, csplit), dim3(64 * G::T), lds, st, in, (const uint16_t*)scratch, two_pass ? partial : out, C,
                     cper, nrows);
  if (two_pass)
    hipLaunchKernelGGL(k_sum_partials, dim3((G::O * G::O + 255) / 256, B), dim3(256), 0, st, partial, out, csplit, G::O * G::O);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: %s", hipGetErrorString(e));
    return 2;
  }
  return 0;
}

template <int HIN, int KH, bool IN_PER_C, bool K_PER_C, bool SUM>
int launch(const void* in, int in_f32, const void* kern, int kern_f32, int split, float* out, void* scratch, int B, int C,
           hipStream_t st) {
#define SRL_X(IF, KF, SP) return launch2<HIN, KH, IN_PER_C, K_PER_C, SUM, IF, KF, SP>(in, kern, out, scratch, B, C, st)
  if (split) {
    if (in_f32 && kern_f32) SRL_X(true, true, true);
    snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: precision 1 (bf16x3) takes float32 operands");
    return 1;
  }
  if (in_f32) { if (kern_f32) SRL_X(true, true, false); else SRL_X(true, false, false); }
  if (kern_f32) SRL_X(false, true, false);
  SRL_X(false, false, false);
#undef SRL_X
}

// (map side, kernel side) of the three modes for the forward shape (H, kh)
bool shapes(int mode, int H, int kh, int* hin, int* ks) {
  const int O = H - kh + 1;
  if (mode == 0) { *hin = H; *ks = kh; }
  else if (mode == 1) { *hin = O + 2 * (kh - 1); *ks = kh; }
  else if (mode == 2) { *hin = H; *ks = O; }
  else return false;
  return (H == 128 && kh == 32) || (H == 64 && kh == 16);
}

}  // namespace

extern "C" {

const char* srl_xcorr_mfma_last_error(void) { return x_err; }

int64_t srl_xcorr_mfma_scratch_bytes(int32_t mode, int32_t precision, int32_t B, int32_t C, int32_t H, int32_t kh) {
  int hin, ks;
  if (B < 1 || C < 1 || !shapes(mode, H, kh, &hin, &ks)) return -1;
  const int64_t KB = 1 + (ks - 1 + 15) / 16;
  const int64_t nrows = (int64_t)(mode == 2 ? B : B * C) * ks;
  int cper, csplit;
  channel_split(B, C, &cper, &csplit);
  const int64_t O = hin - ks + 1;
  const int64_t partials = (mode == 0 && csplit > 1) ? (int64_t)B * csplit * O * O * (int64_t)sizeof(float) : 0;
  return nrows * KB * 64 * 4 * (int64_t)sizeof(uint16_t) * (precision ? 2 : 1) + partials;
}

int srl_xcorr_mfma(int32_t mode, int32_t precision, const void* in, int32_t in_f32, const void* kern, int32_t kern_f32,
                   float* out, void* scratch, int64_t scratch_bytes, int32_t B, int32_t C, int32_t H, int32_t kh,
                   void* stream) {
  int hin, ks;
  if (!in || !kern || !out || !scratch || B < 1 || C < 1 || precision < 0 || precision > 1) {
    snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: bad arguments");
    return 1;
  }
  if (!shapes(mode, H, kh, &hin, &ks)) {
    snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: unsupported mode %d / shape %d, %d (128 / 32 and 64 / 16 are built)", mode, H, kh);
    return 1;
  }
  if (scratch_bytes < srl_xcorr_mfma_scratch_bytes(mode, precision, B, C, H, kh)) {
    snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: scratch too small");
    return 1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (H == 128) {
    if (mode == 0) return launch<128, 32, true, true, true>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
    if (mode == 1) return launch<159, 32, false, true, false>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
    return launch<128, 97, true, false, false>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
  }
  if (mode == 0) return launch<64, 16, true, true, true>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
  if (mode == 1) return launch<79, 16, false, true, false>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
  return launch<64, 49, true, false, false>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
}

}  // extern "C"
