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
// (round 3: 2 k-blocks of 32 for KH = 32, v_mfma_f32_16x16x32_bf16 at full rate; fp32 accumulation).  T does not depend
// on the tile (shift invariance): lane l of a Toeplitz fragment holds kern[i][t0 .. t0 + 7], t0 = 32 j + 8 (l / 16) - l % 16
// — eight CONSECUTIVE elements of the kernel row.  So the fragments are not materialised (rounds 1-2 had a prep kernel write them to global memory: 805 MB
// written and read again per 512 samples of the rollout): the kernel rows of the current (sample, channel) sit in LDS,
// zero-padded on both sides, and a fragment is five aligned 32-bit LDS reads + four v_alignbyte_b32.
//
// Main kernel: one workgroup per sample, one wave per row of output tiles.  One channel of the map is staged in LDS at
// a time (bf16, zero-padded) together with that channel's kernel rows; per kernel row a wave reads its A fragments
// (8 bytes per lane, shared by its tiles: tile t uses k-blocks t .. t + KB - 1), builds the KB Toeplitz fragments of the
// NEXT row from LDS and issues T * KB MFMAs.  Accumulators stay in registers over all kernel rows (and channels, when
// summed).
//
// Precision: 0 = operands rounded to bf16 (products exact in fp32, fp32 accumulation) — the rollout path under bf16
// autocast; 1 = "bf16x3": every fp32 operand is split into hi + lo bf16 parts and hi*hi + hi*lo + lo*hi is
// accumulated (the dropped terms are below 2^-16 relative per product), which is fp32-class accuracy at a third of the
// bf16 MFMA rate and still an order of magnitude above the fp32 vector rate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/stackrl_qnet.h"
#include "srl_bf16.h"

namespace {

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int HIN, int KH>
struct XcorrCfg {
  static constexpr int O = HIN - KH + 1;             // outputs per side
  static constexpr int T = (O + 15) / 16;            // 16 x 16 output tiles per side = waves per workgroup
  // a tile needs k in [0, 16 + KH - 1): KB blocks of 32 (v_mfma_f32_16x16x32_bf16: the K = 16 form runs at half rate on
  // gfx950).  Tile t's block j starts at element 16 (t + 2 j): the A fragments of a row of tiles are the 32-wide windows
  // at 16 u, u = 0 .. NA - 1 (neighbouring windows overlap by half; tile t + 2 reuses tile t's second window).
  static constexpr int KB = (16 + KH - 1 + 31) / 32;
  static constexpr int NA = T + 2 * KB - 2;          // A windows a row of tiles touches
  static constexpr int ROWS = 16 * T + KH - 1;       // staged rows; rows >= HIN stay zero
  static constexpr int RS = 16 * (NA + 1) + 8;       // row stride in elements: windows stay 16-byte aligned
  // kernel rows in LDS: 16 zeros, the KH elements, zeros up to what the last fragment's five-word read touches
  static constexpr int KR = 16 + 32 * KB + 16;       // elements per row (even: rows are 4-byte aligned)
};

__device__ __forceinline__ uint32_t bf16_rne(float f) { return srl_bf16(f); }   // round to nearest even
__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

template <bool F32>
__device__ __forceinline__ float load_elem(const void* p, size_t k) {
  if (F32) return ((const float*)p)[k];
  return bf16_to_f32(((const uint16_t*)p)[k]);
}

// Toeplitz fragment (B operand of v_mfma_f32_16x16x32_bf16: lane l holds column l % 16, rows 8 (l / 16) .. + 7 of
// k-block j) of a kernel row held in LDS with 16 leading zeros: T[k][x] = kern[k - x], i.e. the eight consecutive
// elements 16 + t0 .. 16 + t0 + 7, t0 = 32 j + 8 (l / 16) - l % 16 — from the five aligned words that hold them, shifted
// by one element when the start is odd.
__device__ __forceinline__ bf16x8 toeplitz_frag(const uint16_t* krow, int j, int lane) {
  const int base = 16 + 32 * j + 8 * (lane >> 4) - (lane & 15);
  const uint32_t* w = (const uint32_t*)krow + (base >> 1);
  const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
  const uint32_t sh = (uint32_t)(base & 1) * 2u;
  union { uint32_t u[4]; bf16x8 v; } r;
  r.u[0] = __builtin_amdgcn_alignbyte(w1, w0, sh); r.u[1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
  r.u[2] = __builtin_amdgcn_alignbyte(w3, w2, sh); r.u[3] = __builtin_amdgcn_alignbyte(w4, w3, sh);
  return r.v;
}

// IN_PER_C: the map is per channel ([B][C][HIN][HIN]) or shared by the channels ([B][HIN][HIN]);
// K_PER_C: likewise for the kernel rows; SUM: one output per sample (channels summed) or one per (sample, channel).
// TH: rows of output tiles per workgroup (= waves per workgroup).  A sample's T tile rows are spread over ceil(T / TH)
// workgroups (blockIdx.z), each staging only the 16 TH + KH - 1 map rows it needs: with TH = 4 the forward's workgroup
// holds 76 KB of LDS instead of 108, two fit a CU, and one's staging (global loads, conversion, barriers) runs under the
// other's matrix products — with one workgroup per CU nothing covered it (MFMA pipes busy 58 % of the kernel's time).
template <int HIN, int KH, int TH, bool IN_PER_C, bool K_PER_C, bool SUM, bool F32, bool KF32, bool SPLIT>
__global__ void __launch_bounds__((64 * TH))
k_xcorr_mfma(const void* __restrict__ in, const void* __restrict__ kern, float* __restrict__ out, int C, int cper) {
  typedef XcorrCfg<HIN, KH> G;
  extern __shared__ uint16_t xs[];   // [ROWS_L][RS] bf16 hi parts (+ the same again for the lo parts), then the kernel rows
  constexpr int ROWS_L = 16 * TH + KH - 1;   // staged map rows: local row r = map row 16 TH blockIdx.z + r
  constexpr int TILE = ROWS_L * G::RS;
  constexpr int KTILE = KH * G::KR;
  static_assert(TILE % 8 == 0 && KTILE % 2 == 0 && G::RS % 8 == 0, "A windows are 16-byte LDS reads; kernel rows stay 4-byte aligned");
  uint16_t* ks = xs + (SPLIT ? 2 : 1) * TILE;   // [KH][KR] hi (+ [KH][KR] lo)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int row0 = 16 * TH * blockIdx.z;                 // first map row of this workgroup's tile rows
  const int yt = (tid >> 6) + TH * blockIdx.z;           // this wave's row of output tiles (may lie past the last one)
  constexpr int NT = 64 * TH;
  {   // zero the tile(s) and the kernel rows once: the padding is never written again
    uint32_t* z = (uint32_t*)xs;
    for (int k = tid; k < (SPLIT ? 2 : 1) * (TILE + KTILE) / 2; k += NT) z[k] = 0u;
  }
  f32x4 acc[G::T];
#pragma unroll
  for (int t = 0; t < G::T; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const int arow = 16 * (tid >> 6) + (lane & 15), acol = 8 * (lane >> 4);
  // blockIdx.y takes channels [c0, c1): with few samples the channels are spread over workgroups (summed outputs
  // then land in per-workgroup partials, reduced by k_sum_partials in a fixed order)
  const int c0 = blockIdx.y * cper, c1 = min(C, c0 + cper);
  // PF (the forward with float32 operands per channel — the rollout and the update's forward): a channel's map rows and
  // kernel rows are requested as whole batches of 16-byte loads into registers while the PREVIOUS channel's matrix products
  // run, and converted into LDS after them.  As a plain loop per channel the compiler waited for every load before
  // issuing the next: twelve + four HBM round trips in a row per channel, ~19 us against 9 us of MFMAs (MFMA pipes busy
  // 57 % of the kernel's time with two workgroups per CU).
  constexpr bool PF = IN_PER_C && K_PER_C && F32 && KF32 && HIN % 4 == 0 && (KH * KH) % (4 * NT) == 0;
  constexpr int MAP_IT = PF ? (ROWS_L * (HIN / 4) + NT - 1) / NT : 1, K_IT = PF ? KH * KH / (4 * NT) : 1;
  float4 mreg[MAP_IT], kreg[K_IT];
  const int nrow_pf = min(ROWS_L, HIN - row0);
  auto fetch = [&](int c) {
    const float4* mp = (const float4*)((const float*)in + ((size_t)b * C + c) * HIN * HIN) + row0 * (HIN / 4);
    const float4* kp = (const float4*)((const float*)kern + ((size_t)b * C + c) * KH * KH);
#pragma unroll
    for (int i = 0; i < MAP_IT; ++i) {
      const int k = tid + NT * i;
      mreg[i] = mp[k < nrow_pf * (HIN / 4) ? k : 0];
    }
#pragma unroll
    for (int i = 0; i < K_IT; ++i) kreg[i] = kp[tid + NT * i];
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < MAP_IT; ++i) {
      const int k = tid + NT * i;
      if (k < nrow_pf * (HIN / 4)) {
        const int r = k / (HIN / 4), cc = (k - r * (HIN / 4)) * 4;
        uint32_t h0, l0, h1, l1;
        if (SPLIT) { srl_split_bf16(mreg[i].x, mreg[i].y, h0, l0); srl_split_bf16(mreg[i].z, mreg[i].w, h1, l1); }
        else { h0 = srl_pk_bf16(mreg[i].x, mreg[i].y); h1 = srl_pk_bf16(mreg[i].z, mreg[i].w); l0 = l1 = 0u; }
        *(uint2*)(xs + r * G::RS + cc) = make_uint2(h0, h1);
        if (SPLIT) *(uint2*)(xs + TILE + r * G::RS + cc) = make_uint2(l0, l1);
      }
    }
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const int k4 = 4 * (tid + NT * i), r = k4 / KH, t = k4 - r * KH;   // four consecutive elements of one kernel row
      uint32_t h0, l0, h1, l1;
      if (SPLIT) { srl_split_bf16(kreg[i].x, kreg[i].y, h0, l0); srl_split_bf16(kreg[i].z, kreg[i].w, h1, l1); }
      else { h0 = srl_pk_bf16(kreg[i].x, kreg[i].y); h1 = srl_pk_bf16(kreg[i].z, kreg[i].w); l0 = l1 = 0u; }
      *(uint2*)(ks + r * G::KR + 16 + t) = make_uint2(h0, h1);
      if (SPLIT) *(uint2*)(ks + KTILE + r * G::KR + 16 + t) = make_uint2(l0, l1);
    }
  };
  if (PF && c0 < c1) fetch(c0);
  for (int c = c0; c < c1; ++c) {
    if (PF) {
      __syncthreads();   // the previous channel's reads are done (first trip: the zero fill is complete)
      store();
      __syncthreads();
      if (c + 1 < c1) fetch(c + 1);
    }
    if (!PF && (IN_PER_C || c == c0)) {
      __syncthreads();   // the previous channel's reads are done (first trip: the zero fill is complete)
      const size_t base = (IN_PER_C ? (size_t)b * C + c : (size_t)b) * HIN * HIN;
      if (HIN % 4 == 0) {   // 4 elements per thread and step
        const int nrow = min(ROWS_L, HIN - row0);       // map rows of this workgroup that exist (the others stay zero)
        for (int k = tid; k < nrow * (HIN / 4); k += NT) {
          const int r = k / (HIN / 4), cc = (k - r * (HIN / 4)) * 4;
          const int kg = k + row0 * (HIN / 4);
          float v[4];
          if (F32) {
            const float4 q = ((const float4*)((const float*)in + base))[kg];
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
          } else {
            const uint2 q = ((const uint2*)((const uint16_t*)in + base))[kg];
            v[0] = bf16_to_f32(q.x & 0xffffu); v[1] = bf16_to_f32(q.x >> 16);
            v[2] = bf16_to_f32(q.y & 0xffffu); v[3] = bf16_to_f32(q.y >> 16);
          }
          uint32_t hi[2], lo[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (SPLIT) srl_split_bf16(v[2 * e], v[2 * e + 1], hi[e], lo[e]);
            else hi[e] = srl_pk_bf16(v[2 * e], v[2 * e + 1]);
          }
          *(uint2*)(xs + r * G::RS + cc) = make_uint2(hi[0], hi[1]);
          if (SPLIT) *(uint2*)(xs + TILE + r * G::RS + cc) = make_uint2(lo[0], lo[1]);
        }
      } else {   // odd sides (the padded gradient map): element by element
        const int nrow = min(ROWS_L, HIN - row0);
        for (int k = tid; k < nrow * HIN; k += NT) {
          const int r = k / HIN, cc = k - r * HIN;
          const float v = load_elem<F32>(in, base + (size_t)row0 * HIN + k);
          const uint32_t hi = bf16_rne(v);
          xs[r * G::RS + cc] = (uint16_t)hi;
          if (SPLIT) xs[TILE + r * G::RS + cc] = (uint16_t)bf16_rne(v - bf16_to_f32(hi));
        }
      }
      __syncthreads();
    }
    if (!PF && (K_PER_C || c == c0)) {   // this channel's (or the sample's) kernel rows into LDS, behind 16 zeros each
      if (IN_PER_C || c == c0) {} else __syncthreads();   // (the previous channel's fragment reads are done)
      const size_t kbase = (K_PER_C ? (size_t)b * C + c : (size_t)b) * KH * KH;
      for (int k = tid; k < KH * KH; k += NT) {
        const int r = k / KH, t = k - r * KH;
        const float v = load_elem<KF32>(kern, kbase + k);
        const uint32_t hi = bf16_rne(v);
        ks[r * G::KR + 16 + t] = (uint16_t)hi;
        if (SPLIT) ks[KTILE + r * G::KR + 16 + t] = (uint16_t)bf16_rne(v - bf16_to_f32(hi));
      }
      __syncthreads();
    }
    // kernel-row loop, software-pipelined: the Toeplitz fragments of row i + 1 are built (LDS) before the MFMAs of row i
    // are issued, so their latency is covered by the 21 MFMAs instead of being waited for
    bf16x8 tf[G::KB], tl[SPLIT ? G::KB : 1];
#pragma unroll
    for (int j = 0; j < G::KB; ++j) {
      tf[j] = toeplitz_frag(ks, j, lane);
      if (SPLIT) tl[j] = toeplitz_frag(ks + KTILE, j, lane);
    }
#pragma unroll 2
    for (int i = 0; i < KH; ++i) {
      const int in = i + 1 < KH ? i + 1 : i;
      bf16x8 nf[G::KB], nl[SPLIT ? G::KB : 1];
#pragma unroll
      for (int j = 0; j < G::KB; ++j) {
        nf[j] = toeplitz_frag(ks + in * G::KR, j, lane);
        if (SPLIT) nl[j] = toeplitz_frag(ks + KTILE + in * G::KR, j, lane);
      }
      bf16x8 af[G::NA], al[SPLIT ? G::NA : 1];
      const uint16_t* ar = xs + (arow + i) * G::RS + acol;
#pragma unroll
      for (int u = 0; u < G::NA; ++u) {
        af[u] = *(const bf16x8*)(ar + 16 * u);
        if (SPLIT) al[u] = *(const bf16x8*)(ar + TILE + 16 * u);
      }
#pragma unroll
      for (int t = 0; t < G::T; ++t)
#pragma unroll
        for (int j = 0; j < G::KB; ++j) {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t + 2 * j], tf[j], acc[t], 0, 0, 0);
          if (SPLIT) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t + 2 * j], tl[j], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t + 2 * j], tf[j], acc[t], 0, 0, 0);
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

template <int HIN, int KH, int TH, bool IN_PER_C, bool K_PER_C, bool SUM, bool IF32, bool KF32, bool SPLIT>
int launch2(const void* in, const void* kern, float* out, void* scratch, int B, int C, hipStream_t st) {
  typedef XcorrCfg<HIN, KH> G;
  const size_t lds = sizeof(uint16_t) * ((16 * TH + KH - 1) * G::RS + KH * G::KR) * (SPLIT ? 2 : 1);
  auto fn = k_xcorr_mfma<HIN, KH, TH, IN_PER_C, K_PER_C, SUM, IF32, KF32, SPLIT>;
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
  float* partial = (float*)scratch;
  const bool two_pass = SUM && csplit > 1;
  hipLaunchKernelGGL(fn, dim3(B, csplit, (G::T + TH - 1) / TH), dim3(64 * TH), lds, st, in, kern, two_pass ? partial : out, C, cper);
  if (two_pass)
    hipLaunchKernelGGL(k_sum_partials, dim3((G::O * G::O + 255) / 256, B), dim3(256), 0, st, partial, out, csplit, G::O * G::O);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: %s", hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// THS / THN: tile rows per workgroup with / without the bf16x3 split (without it a whole sample fits twice per CU anyway)
template <int HIN, int KH, int THS, int THN, bool IN_PER_C, bool K_PER_C, bool SUM>
int launch(const void* in, int in_f32, const void* kern, int kern_f32, int split, float* out, void* scratch, int B, int C,
           hipStream_t st) {
#define SRL_X(IF, KF, SP) return launch2<HIN, KH, (SP ? THS : THN), IN_PER_C, K_PER_C, SUM, IF, KF, SP>(in, kern, out, scratch, B, C, st)
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


// ------------------------------------------------------------------------------------------------------------------------
// Round 5: the FORWARD of large batches (the rollout) as a product per MAP ROW instead of per kernel row — no banded operand.
//
//   out[y][x] = sum_c sum_i Z_c[y + i][x][i],      Z_c[rho][x][i] = sum_t X_c[rho][x + t] * W_c[i][t]
//
// For a fixed map row rho, Z_c[rho] is a plain matrix product: A[m = x][k = t] = X_c[rho][x + t] (a Hankel matrix of the row:
// lane (m, g) holds the 8 consecutive elements from x0 + m + 8 g), B[k = t][n = i] = W_c[i][t] (8 consecutive elements of
// kernel row i: one aligned 16-byte LDS read), K = 32 = the kernel's width: ONE v_mfma_f32_16x16x32_bf16 per (x tile, i tile)
// with every MAC useful — the Toeplitz form above issues two K blocks of 32 for 32 taps (half of T is zero) and a 7 x 7 grid of
// tiles; this form issues 7 x tiles x 2 i tiles per (rho, c) over 128 map rows: 0.57 x the MFMAs, and 32 LDS instructions per
// 42 MFMAs where the Toeplitz form has 38 (of four times the bytes).
// The sum over i runs along a DIAGONAL of (rho, i): a wave sweeps rho = 0 .. 127 and keeps, in registers laid out like the
// accumulator tiles (lane n = i), the partial sums S[x][i] of output row y = rho - i; a step to the next map row moves every
// partial one lane up (DPP row_shr:1 within the 16 lanes of a tile row; lane 15 of i-tile 0 carries into lane 0 of i-tile 1 by
// row_ror:1) and adds the new row's products; what arrives at i = 31 is the finished output row y = rho - 31.  The order of
// every sum is fixed (i ascending, channels in the wave's order, waves in index order): bit-identical on repetition.
// One workgroup = one sample, eight waves; wave w takes the channels w, w + 8 and ALL x tiles; its channels' kernel fragments
// stay in registers for the whole sweep; the map is streamed row by row (one float4 per thread per row, requested two rows
// ahead, double-buffered in LDS as four copies shifted by 0 .. 3 elements, so that a Hankel fragment — 8 consecutive elements
// from ANY start — is two 8-byte reads at their natural alignment: 28 ds_read_b64 of 2 LDS cycles per 42 MFMAs); the eight waves' partial output rows
// meet in LDS and are added in wave order.  87 KB of LDS: one workgroup (two waves per SIMD) per CU.

typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lds_u2;

template <bool SPLIT, bool F32>
__global__ void __launch_bounds__(512)
k_xcorr_rows(const void* __restrict__ in, const void* __restrict__ kern, float* __restrict__ out, int C) {
  constexpr int HIN = 128, KH = 32, O = HIN - KH + 1, T = 7, NW = 8;
  constexpr int XS = 160;                     // staged map row stride in elements: 128 + zeros up to what the last window reads
  constexpr int CMAX = 16, NCP = 4;           // NCP copies of every staged row, copy k shifted by k elements
  constexpr int PL = (SPLIT ? 2 : 1);
  constexpr int CPS = CMAX * XS + 32;         // elements per copy: 64 bytes of padding, so that the four copies a half-wave's
                                              // 8-byte reads touch lie on different quarters of the 64 banks
  constexpr int XPL = NCP * CPS;              // elements per plane (hi or lo) of one buffer
  extern __shared__ uint16_t xs[];
  uint16_t* xr = xs;                                            // [2 buffers][PL][NCP][CMAX][XS]
  float* comb = (float*)(xr + 2 * PL * XPL);                    // [2][NW][16 T]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  {   // zero the staged-row buffers once: their tails (columns 128 ..) are never written again
    uint32_t* z = (uint32_t*)xr;
    for (int k = tid; k < 2 * PL * XPL / 2; k += 512) z[k] = 0u;
  }
  // B fragments of this wave's channels (c = wave, wave + 8), in registers for the whole sweep: lane (n, g) holds
  // W_c[16 it + n][8 g .. 8 g + 7] — straight from global memory, once
  bf16x8 bh[2][2], bl[SPLIT ? 2 : 1][2];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int c = wave + NW * ci;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      union { uint32_t u[4]; bf16x8 v; } h, l;
      h.u[0] = h.u[1] = h.u[2] = h.u[3] = 0u; l.u[0] = l.u[1] = l.u[2] = l.u[3] = 0u;
      if (c < C) {
        const size_t off = (((size_t)b * C + c) * KH + 16 * it + n) * KH + 8 * g;
        if (F32) {
          const float4 q0 = *(const float4*)((const float*)kern + off), q1 = *(const float4*)((const float*)kern + off + 4);
          if (SPLIT) {
            srl_split_bf16(q0.x, q0.y, h.u[0], l.u[0]); srl_split_bf16(q0.z, q0.w, h.u[1], l.u[1]);
            srl_split_bf16(q1.x, q1.y, h.u[2], l.u[2]); srl_split_bf16(q1.z, q1.w, h.u[3], l.u[3]);
          } else {
            h.u[0] = srl_pk_bf16(q0.x, q0.y); h.u[1] = srl_pk_bf16(q0.z, q0.w);
            h.u[2] = srl_pk_bf16(q1.x, q1.y); h.u[3] = srl_pk_bf16(q1.z, q1.w);
          }
        } else {
          const uint4 q = *(const uint4*)((const uint16_t*)kern + off);
          h.u[0] = q.x; h.u[1] = q.y; h.u[2] = q.z; h.u[3] = q.w;
        }
      }
      bh[ci][it] = h.v;
      if (SPLIT) bl[ci][it] = l.v;
    }
  }
  // map rows: thread -> (channel tid / 32, columns 4 (tid % 32) .. + 3)
  const int sc = tid >> 5, scol = (tid & 31) * 4;
  const bool sact = sc < C;
  float4 rowreg = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  auto fetch = [&](int rho) {
    if (!sact) return;
    if (F32) rowreg = *(const float4*)((const float*)in + (((size_t)b * C + sc) * HIN + rho) * HIN + scol);
    else {
      const uint2 q = *(const uint2*)((const uint16_t*)in + (((size_t)b * C + sc) * HIN + rho) * HIN + scol);
      rowreg = make_float4(bf16_to_f32(q.x & 0xffffu), bf16_to_f32(q.x >> 16), bf16_to_f32(q.y & 0xffffu), bf16_to_f32(q.y >> 16));
    }
  };
  // the row goes to LDS NCP times, copy k holding X[j + k] at element j: a fragment (8 consecutive elements from any start s)
  // is then ONE 16-byte read at the 8-byte aligned element s & ~3 of copy s & 3.  A thread has the elements 4 q .. 4 q + 3
  // and takes 4 q + 4 .. 4 q + 6 from the next lane (zeros behind the row's last thread).
  auto stash = [&](int buf) {
    float e[7] = {rowreg.x, rowreg.y, rowreg.z, rowreg.w, 0.0f, 0.0f, 0.0f};
    const float nx = __shfl_down(rowreg.x, 1), ny = __shfl_down(rowreg.y, 1), nz = __shfl_down(rowreg.z, 1);
    if ((tid & 31) != 31) { e[4] = nx; e[5] = ny; e[6] = nz; }
    if (!sact) return;
    uint32_t hi[7], lo[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      hi[k] = bf16_rne(e[k]);
      lo[k] = SPLIT ? bf16_rne(e[k] - bf16_to_f32(hi[k])) : 0u;
    }
    uint16_t* d = xr + buf * PL * XPL + sc * XS + scol;
#pragma unroll
    for (int k = 0; k < NCP; ++k) {
      *(uint2*)(d + k * CPS) = make_uint2(hi[k] | (hi[k + 1] << 16), hi[k + 2] | (hi[k + 3] << 16));
      if (SPLIT) *(uint2*)(d + XPL + k * CPS) = make_uint2(lo[k] | (lo[k + 1] << 16), lo[k + 2] | (lo[k + 3] << 16));
    }
  };
  fetch(0);
  __syncthreads();            // the zero fill is complete
  stash(0);
  fetch(1);
  // S[t][it]: accumulator tiles that ARE the diagonal sums: lane n of i-tile `it` holds the partial sum of output row
  // y = rho - 16 it - n, columns 16 t + 4 g + r.  A step to the next map row moves every partial one lane up (row_shr:1, zero
  // into lane 0: a fresh row) and the row's products are accumulated on top by the matrix cores themselves (the shifted
  // registers are the MFMAs' C operand).  The two i-tiles run as two chains of 16: what leaves lane 15 of i-tile 0 at step rho is
  // the first half (i < 16) of output row rho - 15, what leaves lane 15 of i-tile 1 the second half of row rho - 31.
  f32x4 S[T][2];
#pragma unroll
  for (int t = 0; t < T; ++t) { S[t][0] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; S[t][1] = S[t][0]; }
  // per-lane fragment offset (elements) inside a plane, without channel and tile: tile t starts at s = 16 t + n + 8 g, whose
  // copy (s & 3) and aligned part (s & ~3) = 16 t + ((n + 8 g) & ~3) depend on the lane only up to the immediate 16 t
  const int s_lane = n + 8 * g;
  const int lane_off = (s_lane & 3) * CPS + (s_lane & ~3);
  float* ring = comb + 2 * 2 * NW * (16 * T);          // [16][128]: first halves of the output rows waiting for their second
  __syncthreads();
  for (int rho = 0; rho < HIN; ++rho) {
    const uint16_t* xb = xr + (rho & 1) * PL * XPL + lane_off;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          S[t][it][r] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(S[t][it][r]), 0x111, 0xf, 0xf, true));   // row_shr:1
    // A fragments of the seven tiles of both channels: lane (m = n, g) holds X_c[rho][s .. s + 7] — two 8-byte reads each
    // (ds_read_b64: 256 B per LDS clock; as one ds_read2_b64 the pair runs at half that, so every read goes through a 32-bit LDS
    // address the compiler cannot see through and none is merged with a neighbour 8 or 32 bytes away).  All of a step's reads
    // are requested first; the next row's conversion and LDS writes (the other buffer) run while they arrive.
    bf16x8 ah[2][T], al[SPLIT ? 2 : 1][SPLIT ? T : 1];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
      const int c = min(wave + NW * ci, C - 1);
      const uint16_t* pc = xb + c * XS;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        uint32_t pa = (uint32_t)(uintptr_t)(const lds_u16*)(pc + 16 * t);
        asm volatile("" : "+v"(pa));
        uint32_t pb = pa + 8u;
        asm volatile("" : "+v"(pb));
        union { u32x2 h[2]; bf16x8 v; } fh, fl;
        fh.h[0] = *(const lds_u2*)(uintptr_t)pa; fh.h[1] = *(const lds_u2*)(uintptr_t)pb;
        ah[ci][t] = fh.v;
        if (SPLIT) {
          fl.h[0] = *(const lds_u2*)(uintptr_t)(pa + 2u * XPL); fl.h[1] = *(const lds_u2*)(uintptr_t)(pb + 2u * XPL);
          al[ci][t] = fl.v;
        }
      }
    }
    if (rho + 1 < HIN) stash((rho + 1) & 1);
    if (rho + 2 < HIN) fetch(rho + 2);
    // per channel three passes over the fourteen accumulator tiles (hi hi, hi lo, lo hi): an accumulator's next product is
    // fourteen MFMAs away instead of next in line.  (A channel past the last one has zero kernel fragments: its products add 0.)
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
      if (wave + NW * ci < C) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int it = 0; it < 2; ++it) S[t][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ci][t], bh[ci][it], S[t][it], 0, 0, 0);
        if (SPLIT) {
#pragma unroll
          for (int t = 0; t < T; ++t)
#pragma unroll
            for (int it = 0; it < 2; ++it) S[t][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ci][t], bl[ci][it], S[t][it], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < T; ++t)
#pragma unroll
            for (int it = 0; it < 2; ++it) S[t][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ci][t], bh[ci][it], S[t][it], 0, 0, 0);
        }
      }
    }
    // this wave's parts of the two output rows that complete a chain at this step (lanes n = 15)
    if (n == 15) {
      float* cp = comb + ((rho & 1) * 2 * NW + wave) * (16 * T) + 4 * g;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        *(float4*)(cp + 16 * t) = make_float4(S[t][0][0], S[t][0][1], S[t][0][2], S[t][0][3]);
        *(float4*)(cp + NW * 16 * T + 16 * t) = make_float4(S[t][1][0], S[t][1][1], S[t][1][2], S[t][1][3]);
      }
    }
    // (interleaving the row parts' LDS writes and the lane shift with the last pass's matrix products, tile by tile and
    // branch-free, was measured: 665 against 590 us per 512 samples)
    __syncthreads();
    if (tid < O) {      // the waves' parts added in wave order; column tid is always this thread's: the ring needs no barrier
      const float* cp = comb + (rho & 1) * 2 * NW * (16 * T) + tid;
      float p0 = cp[0], p1 = cp[NW * 16 * T];
#pragma unroll
      for (int w = 1; w < NW; ++w) { p0 += cp[w * 16 * T]; p1 += cp[(NW + w) * 16 * T]; }
      float* rp = ring + (rho & 15) * 128 + tid;
      if (rho >= KH - 1) out[((size_t)b * O + (rho - (KH - 1))) * O + tid] = *rp + p1;   // row rho - 31: its first half left at step rho - 16
      *rp = p0;                                                                            // first half of row rho - 15
    }
  }
}

// The same sweep with FOUR waves per sample and two samples per CU: a wave takes four channels (168 MFMAs per map row between
// barriers instead of 84), the staged row is single-buffered (two barriers per row), and the two workgroups that share a CU's
// SIMDs are not in step with one another — one's conversions, LDS traffic and barriers run under the other's matrix products
// (with eight waves of ONE sample per CU every wave is in the same phase at the same time: MFMA pipes busy 43 % of the time).
template <bool SPLIT, bool F32>
__global__ void __launch_bounds__(256, 2)
k_xcorr_rows4(const void* __restrict__ in, const void* __restrict__ kern, float* __restrict__ out, int C) {
  constexpr int HIN = 128, KH = 32, O = HIN - KH + 1, T = 7, NW = 4, CPW = 4;
  constexpr int XS = 160, CMAX = 16, NCP = 4;
  constexpr int PL = (SPLIT ? 2 : 1);
  constexpr int CPS = CMAX * XS + 32;
  constexpr int XPL = NCP * CPS;
  extern __shared__ uint16_t xs[];
  uint16_t* xr = xs;                                            // [PL][NCP][CMAX][XS]
  float* comb = (float*)(xr + PL * XPL);                        // [2 halves][NW][16 T]
  float* ring = comb + 2 * NW * (16 * T);                       // [16][128]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  {
    uint32_t* z = (uint32_t*)xr;
    for (int k = tid; k < PL * XPL / 2; k += 256) z[k] = 0u;
  }
  bf16x8 bh[CPW][2], bl[SPLIT ? CPW : 1][2];
#pragma unroll
  for (int ci = 0; ci < CPW; ++ci) {
    const int c = wave + NW * ci;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      union { uint32_t u[4]; bf16x8 v; } h, l;
      h.u[0] = h.u[1] = h.u[2] = h.u[3] = 0u; l.u[0] = l.u[1] = l.u[2] = l.u[3] = 0u;
      if (c < C) {
        const size_t off = (((size_t)b * C + c) * KH + 16 * it + n) * KH + 8 * g;
        if (F32) {
          const float4 q0 = *(const float4*)((const float*)kern + off), q1 = *(const float4*)((const float*)kern + off + 4);
          if (SPLIT) {
            srl_split_bf16(q0.x, q0.y, h.u[0], l.u[0]); srl_split_bf16(q0.z, q0.w, h.u[1], l.u[1]);
            srl_split_bf16(q1.x, q1.y, h.u[2], l.u[2]); srl_split_bf16(q1.z, q1.w, h.u[3], l.u[3]);
          } else {
            h.u[0] = srl_pk_bf16(q0.x, q0.y); h.u[1] = srl_pk_bf16(q0.z, q0.w);
            h.u[2] = srl_pk_bf16(q1.x, q1.y); h.u[3] = srl_pk_bf16(q1.z, q1.w);
          }
        } else {
          const uint4 q = *(const uint4*)((const uint16_t*)kern + off);
          h.u[0] = q.x; h.u[1] = q.y; h.u[2] = q.z; h.u[3] = q.w;
        }
      }
      bh[ci][it] = h.v;
      if (SPLIT) bl[ci][it] = l.v;
    }
  }
  // map rows: thread -> (channels tid / 32 and tid / 32 + 8, columns 4 (tid % 32) .. + 3)
  const int sc = tid >> 5, scol = (tid & 31) * 4;
  float4 rowreg[2];
  auto fetch = [&](int rho) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = sc + 8 * h;
      rowreg[h] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (c < C) {
        if (F32) rowreg[h] = *(const float4*)((const float*)in + (((size_t)b * C + c) * HIN + rho) * HIN + scol);
        else {
          const uint2 q = *(const uint2*)((const uint16_t*)in + (((size_t)b * C + c) * HIN + rho) * HIN + scol);
          rowreg[h] = make_float4(bf16_to_f32(q.x & 0xffffu), bf16_to_f32(q.x >> 16), bf16_to_f32(q.y & 0xffffu), bf16_to_f32(q.y >> 16));
        }
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = sc + 8 * h;
      float e[7] = {rowreg[h].x, rowreg[h].y, rowreg[h].z, rowreg[h].w, 0.0f, 0.0f, 0.0f};
      const float nx = __shfl_down(rowreg[h].x, 1), ny = __shfl_down(rowreg[h].y, 1), nz = __shfl_down(rowreg[h].z, 1);
      if ((tid & 31) != 31) { e[4] = nx; e[5] = ny; e[6] = nz; }
      if (c < C) {
        uint32_t hi[7], lo[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          hi[k] = bf16_rne(e[k]);
          lo[k] = SPLIT ? bf16_rne(e[k] - bf16_to_f32(hi[k])) : 0u;
        }
        uint16_t* d = xr + c * XS + scol;
#pragma unroll
        for (int k = 0; k < NCP; ++k) {
          *(uint2*)(d + k * CPS) = make_uint2(hi[k] | (hi[k + 1] << 16), hi[k + 2] | (hi[k + 3] << 16));
          if (SPLIT) *(uint2*)(d + XPL + k * CPS) = make_uint2(lo[k] | (lo[k + 1] << 16), lo[k + 2] | (lo[k + 3] << 16));
        }
      }
    }
  };
  fetch(0);
  __syncthreads();
  stash();
  fetch(1);
  f32x4 S[T][2];
#pragma unroll
  for (int t = 0; t < T; ++t) { S[t][0] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; S[t][1] = S[t][0]; }
  const int s_lane = n + 8 * g;
  const uint16_t* xb = xr + (s_lane & 3) * CPS + (s_lane & ~3);
  __syncthreads();
  for (int rho = 0; rho < HIN; ++rho) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          S[t][it][r] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(S[t][it][r]), 0x111, 0xf, 0xf, true));   // row_shr:1
#pragma unroll
    for (int ci = 0; ci < CPW; ++ci) {
      const int c = wave + NW * ci;
      if (c < C) {
        const uint16_t* pc = xb + c * XS;
        bf16x8 ah[T], al[SPLIT ? T : 1];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          uint32_t pa = (uint32_t)(uintptr_t)(const lds_u16*)(pc + 16 * t);
          asm volatile("" : "+v"(pa));
          uint32_t pb = pa + 8u;
          asm volatile("" : "+v"(pb));
          union { u32x2 h[2]; bf16x8 v; } fh, fl;
          fh.h[0] = *(const lds_u2*)(uintptr_t)pa; fh.h[1] = *(const lds_u2*)(uintptr_t)pb;
          ah[t] = fh.v;
          if (SPLIT) {
            fl.h[0] = *(const lds_u2*)(uintptr_t)(pa + 2u * XPL); fl.h[1] = *(const lds_u2*)(uintptr_t)(pb + 2u * XPL);
            al[t] = fl.v;
          }
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int it = 0; it < 2; ++it) S[t][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh[ci][it], S[t][it], 0, 0, 0);
        if (SPLIT) {
#pragma unroll
          for (int t = 0; t < T; ++t)
#pragma unroll
            for (int it = 0; it < 2; ++it) S[t][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bl[ci][it], S[t][it], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < T; ++t)
#pragma unroll
            for (int it = 0; it < 2; ++it) S[t][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bh[ci][it], S[t][it], 0, 0, 0);
        }
      }
    }
    if (n == 15) {
      float* cp = comb + wave * (16 * T) + 4 * g;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        *(float4*)(cp + 16 * t) = make_float4(S[t][0][0], S[t][0][1], S[t][0][2], S[t][0][3]);
        *(float4*)(cp + NW * 16 * T + 16 * t) = make_float4(S[t][1][0], S[t][1][1], S[t][1][2], S[t][1][3]);
      }
    }
    __syncthreads();        // every wave has read row rho; the row parts are in LDS
    if (tid < O) {
      const float* cp = comb + tid;
      float p0 = cp[0], p1 = cp[NW * 16 * T];
#pragma unroll
      for (int w = 1; w < NW; ++w) { p0 += cp[w * 16 * T]; p1 += cp[(NW + w) * 16 * T]; }
      float* rp = ring + (rho & 15) * 128 + tid;
      if (rho >= KH - 1) out[((size_t)b * O + (rho - (KH - 1))) * O + tid] = *rp + p1;
      *rp = p0;
    }
    if (rho + 1 < HIN) stash();
    if (rho + 2 < HIN) fetch(rho + 2);
    __syncthreads();        // row rho + 1 is in LDS; the row parts have been consumed
  }
}

constexpr size_t rows4_lds_bytes(bool split) {
  return sizeof(uint16_t) * ((split ? 2 : 1) * 4 * (16 * 160 + 32)) + sizeof(float) * (2 * 4 * 112 + 16 * 128);
}

constexpr size_t rows_lds_bytes(bool split) {
  return sizeof(uint16_t) * (2 * (split ? 2 : 1) * 4 * (16 * 160 + 32)) + sizeof(float) * (2 * 2 * 8 * 112 + 16 * 128);
}

template <bool SPLIT, bool F32>
int launch_rows(const void* in, const void* kern, float* out, int B, int C, hipStream_t st) {
  const char* var = getenv("SRL_XCORR_ROWS_WAVES");      // (A / B: "8" = one eight-wave workgroup per CU)
  if (!(var && var[0] == '8')) {
    auto fn4 = k_xcorr_rows4<SPLIT, F32>;
    const size_t lds4 = rows4_lds_bytes(SPLIT);
    static bool opted4 = false;
    if (!opted4) {
      hipError_t e = hipFuncSetAttribute((const void*)fn4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
      if (e != hipSuccess) {
        snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: LDS opt-in of %zu bytes failed: %s", lds4, hipGetErrorString(e));
        return 2;
      }
      opted4 = true;
    }
    hipLaunchKernelGGL(fn4, dim3(B), dim3(256), lds4, st, in, kern, out, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: %s", hipGetErrorString(e));
      return 2;
    }
    return 0;
  }
  auto fn = k_xcorr_rows<SPLIT, F32>;
  const size_t lds = rows_lds_bytes(SPLIT);
  static bool opted = false;   // per instantiation; set once, outside any later stream capture
  if (!opted) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: LDS opt-in of %zu bytes failed: %s", lds, hipGetErrorString(e));
      return 2;
    }
    opted = true;
  }
  hipLaunchKernelGGL(fn, dim3(B), dim3(512), lds, st, in, kern, out, C);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(x_err, sizeof x_err, "srl_xcorr_mfma: %s", hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// the row-product forward serves batches that fill the chip with one workgroup per sample (the rollout); smaller batches (the
// update's 32 / 64 samples) keep the Toeplitz kernel with its channel split.  SRL_XCORR_ROWS=0 / 1 forces either (tests, A / B).
bool use_rows(int mode, int precision, int in_f32, int kern_f32, int B, int C, int H, int kh) {
  if (mode != 0 || H != 128 || kh != 32 || C < 1 || C > 16 || in_f32 != kern_f32 || (precision == 1 && !in_f32)) return false;
  const char* force = getenv("SRL_XCORR_ROWS");   // (read at every call: a test switches it)
  if (force && force[0] == '0') return false;
  if (force && force[0] == '1') return true;
  return B >= 192;
}

}  // namespace

extern "C" {

const char* srl_xcorr_mfma_last_error(void) { return x_err; }

int64_t srl_xcorr_mfma_scratch_bytes(int32_t mode, int32_t precision, int32_t B, int32_t C, int32_t H, int32_t kh) {
  int hin, ks;
  if (B < 1 || C < 1 || !shapes(mode, H, kh, &hin, &ks)) return -1;
  int cper, csplit;
  channel_split(B, C, &cper, &csplit);
  const int64_t O = hin - ks + 1;
  // the per-workgroup partial outputs of the forward when the channels are spread over workgroups (small batches); the
  // Toeplitz fragments are no longer materialised
  return (mode == 0 && csplit > 1) ? (int64_t)B * csplit * O * O * (int64_t)sizeof(float) : 0;
}

int srl_xcorr_mfma(int32_t mode, int32_t precision, const void* in, int32_t in_f32, const void* kern, int32_t kern_f32,
                   float* out, void* scratch, int64_t scratch_bytes, int32_t B, int32_t C, int32_t H, int32_t kh,
                   void* stream) {
  int hin, ks;
  if (!in || !kern || !out || B < 1 || C < 1 || precision < 0 || precision > 1 ||
      (!scratch && srl_xcorr_mfma_scratch_bytes(mode, precision, B, C, H, kh) > 0)) {
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
  if (use_rows(mode, precision, in_f32, kern_f32, B, C, H, kh)) {
    if (precision == 1) return launch_rows<true, true>(in, kern, out, B, C, st);
    if (in_f32) return launch_rows<false, true>(in, kern, out, B, C, st);
    return launch_rows<false, false>(in, kern, out, B, C, st);
  }
  if (H == 128) {
    // tile rows per workgroup: 4 of 7 (forward: 76 KB of LDS with the bf16x3 split, two workgroups per CU), 3 of 8 (d/dx:
    // 70 KB), both rows of the d/dw's 2 (its 97 kernel rows are the larger part of the 140 KB)
    if (mode == 0) return launch<128, 32, 4, 7, true, true, true>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
    if (mode == 1) return launch<159, 32, 3, 4, false, true, false>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
    return launch<128, 97, 2, 2, true, false, false>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
  }
  if (mode == 0) return launch<64, 16, 4, 4, true, true, true>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
  if (mode == 1) return launch<79, 16, 4, 4, false, true, false>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
  return launch<64, 49, 1, 1, true, false, false>(in, in_f32, kern, kern_f32, precision, out, scratch, B, C, st);
}

}  // extern "C"
