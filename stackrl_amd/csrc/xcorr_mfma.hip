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
