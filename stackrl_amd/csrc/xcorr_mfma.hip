// xcorr_mfma.hip — K6 on the matrix cores: per-sample VALID cross-correlation of the Q-net (layers.py:21-38) for the
// bf16 rollout path, gfx950.
//
//   out[b, y, x] = sum_{c, i, t} X[b, c, y + i, x + t] * W[b, c, i, t]          X [C, H, H], W [C, KH, KH]
//
// The op has no batch-shared operand (every sample brings its own kernel), so it is not a library GEMM.  It is made
// MFMA-shaped per (channel c, kernel row i): the 1-D correlation along x is a product with a banded Toeplitz matrix,
//
//   out[y, x] += sum_k A[y, k] * T[k, x],     A[y, k] = X[c, y + i, x0 + k],     T[k, x] = W[c, i, k - x]  (0 <= k - x < KH)
//
// For a 16 x 16 output tile at (y0, x0) only k in [0, 16 + KH - 1) matters: 1 + KH/16 k-blocks of 16
// (v_mfma_f32_16x16x16_bf16, fp32 accumulation), i.e. 50 % of the issued multiply-adds are useful at KH = 32.  T does
// not depend on the tile (shift invariance), so its MFMA fragments are built once per (b, c, i) by a small prep kernel
// (k_xcorr_toeplitz, 1.5 KB per kernel row) and then feed every tile of the sample.
//
// Main kernel: one workgroup per sample, one wave per row of output tiles (7 waves for 97 x 97).  One channel of X
// (128 x 128 bf16 = 32 KB, zero-padded to 143 x 148) is staged in LDS at a time; per kernel row a wave reads 9
// A fragments (8 bytes per lane, shared by its 7 tiles: tile t uses k-blocks t .. t + 2), 3 T fragments (global,
// L2-resident: all waves of the sample read the same ones) and issues 21 MFMAs.  Accumulators (7 tiles x 4 VGPRs)
// stay in registers over all (c, i); out is written once, fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stackrl_qnet.h"

namespace {

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int H, int KH>
struct XcorrCfg {
  static constexpr int O = H - KH + 1;          // outputs per side (97 | 49)
  static constexpr int T = (O + 15) / 16;       // 16 x 16 output tiles per side (7 | 4)
  static constexpr int KB = 1 + KH / 16;        // k-blocks per tile and kernel row (3 | 2)
  static constexpr int NKB = T + KH / 16;       // k-blocks a row of tiles touches (9 | 5)
  static constexpr int ROWS = 16 * T + KH - 1;  // staged rows; rows >= H stay zero (143 | 79)
  static constexpr int RS = 16 * NKB + 4;       // row stride in elements (148 | 84): 8-byte aligned rows
};

// Toeplitz fragments: wt[b][c][i][j][lane][r] = W[b][c][i][t], t = 16 j + 4 (lane / 16) + r - lane % 16, zero outside [0, KH)
// (B operand of v_mfma_f32_16x16x16_bf16: lane l holds column l % 16, rows 4 (l / 16) .. + 3 of the k-block)
template <int KH>
__global__ void __launch_bounds__(64 * (1 + KH / 16)) k_xcorr_toeplitz(const uint16_t* __restrict__ w,
                                                                        uint16_t* __restrict__ wt) {
  constexpr int KB = 1 + KH / 16;
  const size_t row = blockIdx.x;   // (b, c, i) flattened
  const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint16_t* wr = w + row * KH;
  uint16_t v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = 16 * j + 4 * (lane >> 4) + r - (lane & 15);
    v[r] = (t >= 0 && t < KH) ? wr[t] : (uint16_t)0;
  }
  uint2 pk = make_uint2((uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16));
  ((uint2*)wt)[(row * KB + j) * 64 + lane] = pk;
}

template <int H, int KH>
__global__ void __launch_bounds__((64 * XcorrCfg<H, KH>::T)) k_xcorr_mfma(const uint16_t* __restrict__ x,
                                                                        const uint16_t* __restrict__ wt,
                                                                        float* __restrict__ out, int C) {
  typedef XcorrCfg<H, KH> G;
  extern __shared__ uint16_t xs[];   // [ROWS][RS] bf16 bits
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, yt = tid >> 6;
  constexpr int NT = 64 * G::T;
  {   // zero the whole tile once: the padding rows / columns are never written again
    uint2* z = (uint2*)xs;
    for (int k = tid; k < G::ROWS * G::RS / 4; k += NT) z[k] = make_uint2(0u, 0u);
  }
  f32x4 acc[G::T];
#pragma unroll
  for (int t = 0; t < G::T; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const int arow = 16 * yt + (lane & 15), acol = 4 * (lane >> 4);
  for (int c = 0; c < C; ++c) {
    __syncthreads();   // the previous channel's reads are done (first trip: the zero fill is complete)
    {   // stage X[b][c]: H x H bf16, 16-byte chunks (8 elements), rows are RS apart in LDS
      const uint4* src = (const uint4*)(x + ((size_t)b * C + c) * H * H);
      for (int k = tid; k < H * H / 8; k += NT) {
        const uint4 v = src[k];
        const int r = k / (H / 8), cc = (k - r * (H / 8)) * 8;
        uint2* d = (uint2*)(xs + r * G::RS + cc);   // RS is a multiple of 4 elements: 8-byte aligned
        d[0] = make_uint2(v.x, v.y); d[1] = make_uint2(v.z, v.w);
      }
    }
    __syncthreads();
    const bf16x4* tw = (const bf16x4*)wt + (((size_t)b * C + c) * KH) * G::KB * 64 + lane;
#pragma unroll 2
    for (int i = 0; i < KH; ++i) {
      bf16x4 tf[G::KB];
#pragma unroll
      for (int j = 0; j < G::KB; ++j) tf[j] = tw[((size_t)i * G::KB + j) * 64];
      bf16x4 af[G::NKB];
      const uint16_t* ar = xs + (arow + i) * G::RS + acol;
#pragma unroll
      for (int kb = 0; kb < G::NKB; ++kb) af[kb] = *(const bf16x4*)(ar + 16 * kb);
#pragma unroll
      for (int t = 0; t < G::T; ++t)
#pragma unroll
        for (int j = 0; j < G::KB; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af[t + j], tf[j], acc[t], 0, 0, 0);
    }
  }
  // D fragment: lane l holds column l % 16, rows 4 (l / 16) .. + 3 of the tile
  float* ob = out + (size_t)b * G::O * G::O;
#pragma unroll
  for (int t = 0; t < G::T; ++t) {
    const int xcol = 16 * t + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = 16 * yt + 4 * (lane >> 4) + r;
      if (y < G::O && xcol < G::O) ob[(size_t)y * G::O + xcol] = acc[t][r];
    }
  }
}

thread_local char x_err[256] = "";

template <int H, int KH>
int launch(const void* x, const void* w, float* out, void* scratch, int B, int C, hipStream_t st) {
  typedef XcorrCfg<H, KH> G;
  hipLaunchKernelGGL((k_xcorr_toeplitz<KH>), dim3((unsigned)((size_t)B * C * KH)), dim3(64 * G::KB), 0, st,
                     (const uint16_t*)w, (uint16_t*)scratch);
  const size_t lds = sizeof(uint16_t) * G::ROWS * G::RS;
  hipLaunchKernelGGL((k_xcorr_mfma<H, KH>), dim3(B), dim3(64 * G::T), lds, st, (const uint16_t*)x,
                     (const uint16_t*)scratch, out, C);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(x_err, sizeof x_err, "srl_xcorr_forward_bf16: %s", hipGetErrorString(e));
    return 2;
  }
  return 0;
}

}  // namespace

extern "C" {

const char* srl_xcorr_bf16_last_error(void) { return x_err; }

int64_t srl_xcorr_bf16_scratch_bytes(int32_t B, int32_t C, int32_t kh) {
  if (B < 1 || C < 1 || (kh != 16 && kh != 32)) return -1;
  return (int64_t)B * C * kh * (1 + kh / 16) * 64 * 4 * (int64_t)sizeof(uint16_t);
}

int srl_xcorr_forward_bf16(const void* x, const void* w, float* out, void* scratch, int64_t scratch_bytes, int32_t B,
                           int32_t C, int32_t H, int32_t W, int32_t kh, int32_t kw, void* stream) {
  if (!x || !w || !out || !scratch || B < 1 || C < 1 || H != W || kh != kw) {
    snprintf(x_err, sizeof x_err, "srl_xcorr_forward_bf16: bad arguments");
    return 1;
  }
  if (scratch_bytes < srl_xcorr_bf16_scratch_bytes(B, C, kh)) {
    snprintf(x_err, sizeof x_err, "srl_xcorr_forward_bf16: scratch too small");
    return 1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (H == 128 && kh == 32) return launch<128, 32>(x, w, out, scratch, B, C, st);
  if (H == 64 && kh == 16) return launch<64, 16>(x, w, out, scratch, B, C, st);
  snprintf(x_err, sizeof x_err, "srl_xcorr_forward_bf16: unsupported shape %dx%d / %dx%d (128/32 and 64/16 are built)", H, W, kh, kw);
  return 1;
}

}  // extern "C"
