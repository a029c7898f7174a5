// train_conv.hip — the convolutions of the DQN minibatch UPDATE (forward with saved activations, data gradient, weight
// gradient) written by hand for gfx950, in true float32 on the matrix cores.
//
// Reference: `DQN.train` (stackrl/agents/dqn.py:397-476) differentiates `DeepQSiamFCN` (stackrl/nets/models.py:106-201,
// `layers.unet` stackrl/nets/layers.py:135-259) with TensorFlow's float32 convolution kernels.  Rounds 1-2 left these to
// MIOpen (igemm fwd / bwd / wrw, atomically accumulating split-K variants with zero-fill launches, NCHW <-> NHWC weight
// copies).  Here every convolution of the update is one of two implicit-GEMM kernels on v_mfma_f32_16x16x4_f32 — float32
// products, float32 accumulation: the reference's dtype, no bf16 split — over float32 channels-last activations:
//
//   k_tconv<TAPS, TW>   y[p][co] = act(bias[co] + sum_{t, ci} x[p + d(t)][ci] w[t][ci][co])
//                       TAPS = 9: 3 x 3, stride 1, SAME.  TAPS = 1: 1 x 1 — which is also the 2 x 2 stride-2 transposed
//                       convolution (`up{i}`, layers.py:222-229) seen as cin -> 4 cout channels followed by a
//                       depth-to-space store, and its data gradient (4 cout -> cin on the space-to-depth gradient).
//                       The data gradient of a 3 x 3 layer is the same kernel on the flipped, transposed weights.
//   k_twrw<TAPS, TW>    gw[t][ci][co] = sum_p x[p + d(t)][ci] gz[p][co]: a GEMM whose reduction runs over the pixels;
//                       every workgroup reduces its pixel tiles in a fixed order and writes ONE partial, a second kernel
//                       adds the partials in index order (no atomics, no zero-fill, bit-identical on repetition).
//   k_tact_bwd          gz = (g [+ the gradient routed back through the 2 x 2 max-pool]) * [y > 0], the bias gradient
//                       as fixed-order partial sums; optionally stored space-to-depth (for the transposed convolutions).
//   k_trepack           all packed weight layouts of all layers from the flat parameter bucket in ONE launch.
//
// Operand layouts are chosen so that no transposition is needed: with D[m][n] += A[m][k] B[k][n] and lane l holding
// A[m = l % 16][k = l / 16], B[k = l / 16][n = l % 16], D[m = 4 (l / 16) + i][n = l % 16] (i = 0..3):
//   conv: m = output channel, n = pixel, k = input channel -> A from packed weights [t][ci][co] (64-byte rows), B from the
//         LDS tile of x (pixel stride CK + 1 words: conflict-free), D = 4 consecutive channels of a pixel (16-byte stores)
//   wrw:  m = input channel, n = output channel, k = pixel -> A from the LDS tile of x, B straight from gz in global
//         memory (64-byte rows), one B fragment serves all nine taps.
// Bound: MFMA (float32: 1/16 of the bf16 rate) for the wide layers, HBM / launch latency for the thin ones; the whole
// update is hidden under the env step in `Training.run`, so the kernels are written for clarity and determinism first.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stackrl_qnet.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

thread_local char t_err[256] = "";

int t_finish(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(t_err, sizeof t_err, "%s: %s", what, hipGetErrorString(e)); return 2; }
  return 0;
}

int t_bad(const char* msg) { snprintf(t_err, sizeof t_err, "%s", msg); return 1; }

constexpr int CK = 16;        // input channels staged per chunk
constexpr int CKP = CK + 1;   // LDS pixel stride in words

// ---------------------------------------------------------------------------------------------------------------- tiles
// A tile is 256 output pixels = 16 n-tiles of 16.  TW = 16: a 16 x 16 patch of one sample (n-tile = a row); TW = 8: an
// 8 x 8 patch of four consecutive samples (n-tile = two rows); TAPS = 1: 256 consecutive pixels of the flattened batch.
template <int TAPS, int TW>
struct Tile {
  int b0, y0, x0;         // first sample, patch origin
  long long p0;           // TAPS == 1: first flattened pixel
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int PW = (TAPS == 9 ? TW + 2 : 0);                     // staged patch width per sample
  static constexpr int NS = TW == 16 ? 1 : 4;                             // samples per tile
  static constexpr int LPIX = TAPS == 9 ? NS * PW * PW : 256;             // staged pixels
  __device__ __forceinline__ void init(long long tile, int B, int H, int W) {
    if (TAPS == 1) { p0 = tile * 256; b0 = y0 = x0 = 0; return; }
    const int tx = (W + TW - 1) / TW, ty = (H + TW - 1) / TW;
    const int per = tx * ty;
    const int g = (int)(tile / per), r = (int)(tile % per);
    b0 = g * NS; y0 = (r / tx) * TW; x0 = (r % tx) * TW; p0 = 0;
  }
  // output pixel q (0..255) of the tile -> sample, row, column (TAPS == 9)
  __device__ __forceinline__ void out_pixel(int q, int& s, int& y, int& x) const {
    if (TW == 16) { s = 0; y = q >> 4; x = q & 15; }
    else { s = q >> 6; y = (q >> 3) & 7; x = q & 7; }
  }
  // LDS pixel index of output pixel q shifted by tap t
  __device__ __forceinline__ int lds_pixel(int q, int t) const {
    if (TAPS == 1) return q;
    int s, y, x; out_pixel(q, s, y, x);
    return (s * PW + y + t / 3) * PW + x + t % 3;
  }
};

// stage channels [c0, c0 + CK) of the tile's input patch (zero outside the map / past cin)
template <int TAPS, int TW>
__device__ __forceinline__ void stage_x(float* lds, const Tile<TAPS, TW>& T, const float* __restrict__ x, int xs, int xo,
                                        int B, int H, int W, int cin, int c0, int ck) {
  typedef Tile<TAPS, TW> TT;
  for (int idx = threadIdx.x; idx < TT::LPIX * ck; idx += 256) {
    const int lp = idx / ck, c = idx - lp * ck;
    float v = 0.0f;
    if (TAPS == 1) {
      const long long p = T.p0 + lp;
      if (p < (long long)B * H * W && c0 + c < cin) v = x[p * xs + xo + c0 + c];
    } else {
      const int s = lp / (TT::PW * TT::PW), r = lp - s * TT::PW * TT::PW;
      const int yy = T.y0 - 1 + r / TT::PW, xx = T.x0 - 1 + r % TT::PW, b = T.b0 + s;
      if (b < B && yy >= 0 && yy < H && xx >= 0 && xx < W && c0 + c < cin)
        v = x[(((long long)b * H + yy) * W + xx) * xs + xo + c0 + c];
    }
    lds[lp * CKP + c] = v;
  }
}

// The same in two halves for tensors whose channel slices are 16-byte aligned (every layer but the thin first ones):
// stage_fetch reads the chunk's 16 channels as float4s into registers — issued BEFORE the matrix products of the chunk in
// flight, so that the global-memory latency hides under them — and stage_put writes them to LDS after the barrier.
template <int TAPS, int TW>
struct StageRegs {
  static constexpr int NV = (Tile<TAPS, TW>::LPIX * (CK / 4) + 255) / 256;
  float4 v[NV];
};

template <int TAPS, int TW>
__device__ __forceinline__ void stage_fetch(StageRegs<TAPS, TW>& R, const Tile<TAPS, TW>& T, const float* __restrict__ x, int xs,
                                            int xo, int B, int H, int W, int cin, int c0) {
  typedef Tile<TAPS, TW> TT;
#pragma unroll
  for (int i = 0; i < StageRegs<TAPS, TW>::NV; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int lp = idx >> 2, c = (idx & 3) * 4;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (lp < TT::LPIX && c0 + c < cin) {
      if (TAPS == 1) {
        const long long p = T.p0 + lp;
        if (p < (long long)B * H * W) v = *(const float4*)(x + p * xs + xo + c0 + c);
      } else {
        const int s = lp / (TT::PW * TT::PW), r = lp - s * TT::PW * TT::PW;
        const int yy = T.y0 - 1 + r / TT::PW, xx = T.x0 - 1 + r % TT::PW, b = T.b0 + s;
        if (b < B && yy >= 0 && yy < H && xx >= 0 && xx < W)
          v = *(const float4*)(x + (((long long)b * H + yy) * W + xx) * xs + xo + c0 + c);
      }
    }
    R.v[i] = v;
  }
}

template <int TAPS, int TW>
__device__ __forceinline__ void stage_put(float* lds, const StageRegs<TAPS, TW>& R) {
  typedef Tile<TAPS, TW> TT;
#pragma unroll
  for (int i = 0; i < StageRegs<TAPS, TW>::NV; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int lp = idx >> 2, c = (idx & 3) * 4;
    if (lp < TT::LPIX) {
      float* d = lds + lp * CKP + c;
      d[0] = R.v[i].x; d[1] = R.v[i].y; d[2] = R.v[i].z; d[3] = R.v[i].w;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- conv
// grid.x = tiles, grid.y = chunks of COT * 16 output channels.  wp: [TAPS][cin_p][cout], cin_p = cin rounded up to 4.
// d2s > 0: output channel k = q * d2s + co is stored at pixel (2 y + q / 2, 2 x + q % 2), channel co of a map twice the
// size (the transposed convolution's depth-to-space), bias indexed by co.
template <int TAPS, int TW, int COT>
__global__ void __launch_bounds__(256) k_tconv(const float* __restrict__ x, int xs, int xo, const float* __restrict__ wp,
                                               const float* __restrict__ bias, float* __restrict__ y, int ys, int yo,
                                               int B, int H, int W, int cin, int cout, int relu, int d2s) {
  typedef Tile<TAPS, TW> TT;
  __shared__ float lds[TT::LPIX * CKP];
  TT T; T.init(blockIdx.x, B, H, W);
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int co0 = blockIdx.y * COT * 16;
  const int cin_p = (cin + 3) & ~3;
  floatx4 acc[COT][4];
#pragma unroll
  for (int ct = 0; ct < COT; ++ct)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[ct][nt] = (floatx4){0.0f, 0.0f, 0.0f, 0.0f};
  int lp[4][TAPS];      // LDS pixel of this lane's column in each of the wave's four n-tiles, per tap
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) lp[nt][t] = T.lds_pixel((4 * wv + nt) * 16 + (l & 15), t) * CKP + (l >> 4);
  // aligned: every channel slice starts on a 16-byte boundary and cin is a multiple of 16 -> float4 staging, the next
  // chunk's global loads issued before the current chunk's matrix products
  const bool aligned = (xs & 3) == 0 && (xo & 3) == 0 && (cin & 15) == 0;
  StageRegs<TAPS, TW> R;
  if (aligned) stage_fetch<TAPS, TW>(R, T, x, xs, xo, B, H, W, cin, 0);
  for (int c0 = 0; c0 < cin_p; c0 += CK) {
    const int ck = cin_p - c0 < CK ? cin_p - c0 : CK;
    __syncthreads();
    if (aligned) stage_put<TAPS, TW>(lds, R);
    else stage_x<TAPS, TW>(lds, T, x, xs, xo, B, H, W, cin, c0, ck);
    __syncthreads();
    if (aligned && c0 + CK < cin_p) stage_fetch<TAPS, TW>(R, T, x, xs, xo, B, H, W, cin, c0 + CK);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const float* wt = wp + ((long long)t * cin_p + c0 + (l >> 4)) * cout + co0 + (l & 15);
      if (ck == CK) {
        float a[4][COT];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int ct = 0; ct < COT; ++ct) a[kk][ct] = co0 + ct * 16 < cout ? wt[(long long)kk * 4 * cout + ct * 16] : 0.0f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          float b[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) b[nt] = lds[lp[nt][t] + 4 * kk];
#pragma unroll
          for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][ct], b[nt], acc[ct][nt], 0, 0, 0);
        }
      } else {
        for (int k4 = 0; k4 < ck; k4 += 4) {
          float a[COT], b[4];
#pragma unroll
          for (int ct = 0; ct < COT; ++ct) a[ct] = co0 + ct * 16 < cout ? wt[(long long)k4 * cout + ct * 16] : 0.0f;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) b[nt] = lds[lp[nt][t] + k4];
#pragma unroll
          for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ct], b[nt], acc[ct][nt], 0, 0, 0);
        }
      }
    }
  }
  // epilogue: lane holds channels co0 + 16 ct + 4 (l / 16) + i of pixel (n-tile, l % 16)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int q = (4 * wv + nt) * 16 + (l & 15);
    long long pix; bool ok;
    int py = 0, px = 0, pb = 0;
    if (TAPS == 1) {
      const long long p = T.p0 + q;
      ok = p < (long long)B * H * W; pix = p;
      if (d2s) { px = (int)(p % W); const long long r = p / W; py = (int)(r % H); pb = (int)(r / H); }
    } else {
      int s, yy, xx; T.out_pixel(q, s, yy, xx);
      pb = T.b0 + s; py = T.y0 + yy; px = T.x0 + xx;
      ok = pb < B && py < H && px < W;
      pix = ((long long)pb * H + py) * W + px;
    }
    if (!ok) continue;
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
      const int k = co0 + ct * 16 + 4 * (l >> 4);
      if (k >= cout) continue;
      int co = k; long long op = pix;
      if (d2s) { const int qd = k / d2s; co = k - qd * d2s; op = ((long long)pb * 2 * H + 2 * py + (qd >> 1)) * 2 * W + 2 * px + (qd & 1); }
      floatx4 v = acc[ct][nt];
      if (bias) { v[0] += bias[co]; v[1] += bias[co + 1]; v[2] += bias[co + 2]; v[3] += bias[co + 3]; }
      if (relu) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f); }
      *(float4*)(y + op * ys + yo + co) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- wrw
// grid.x = pixel-tile groups G, grid.y = input-channel chunks of 16, grid.z = output-channel chunks of COT * 16.
// partial: [G][TAPS][cin_p16][cout] with cin_p16 = cin rounded up to 16.  gz contiguous [pixels][cout].
template <int TAPS, int TW, int COT>
__global__ void __launch_bounds__(256) k_twrw(const float* __restrict__ x, int xs, int xo, const float* __restrict__ gz,
                                              float* __restrict__ partial, int B, int H, int W, int cin, int cout,
                                              long long ntiles) {
  typedef Tile<TAPS, TW> TT;
  constexpr int LDSW = TT::LPIX * CKP > 4 * 64 * 4 ? TT::LPIX * CKP : 4 * 64 * 4;
  __shared__ float lds[LDSW];
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * COT * 16;
  const int cin16 = (cin + 15) & ~15;
  floatx4 acc[TAPS][COT];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) acc[t][ct] = (floatx4){0.0f, 0.0f, 0.0f, 0.0f};
  const long long G = gridDim.x;
  const long long per = (ntiles + G - 1) / G;
  long long t0 = blockIdx.x * per, t1 = t0 + per;
  if (t1 > ntiles) t1 = ntiles;
  for (long long tile = t0; tile < t1; ++tile) {
    TT T; T.init(tile, B, H, W);
    __syncthreads();
    stage_x<TAPS, TW>(lds, T, x, xs, xo, B, H, W, cin, ci0, CK);
    __syncthreads();
    // the wave's 64 pixels in 16 k-steps of 4: pixel q = 64 wv + 4 ks + (l / 16)
#pragma unroll 2
    for (int ks = 0; ks < 16; ++ks) {
      const int q = 64 * wv + 4 * ks + (l >> 4);
      long long pix; bool ok;
      if (TAPS == 1) { pix = T.p0 + q; ok = pix < (long long)B * H * W; }
      else {
        int s, yy, xx; T.out_pixel(q, s, yy, xx);
        const int pb = T.b0 + s, py = T.y0 + yy, px = T.x0 + xx;
        ok = pb < B && py < H && px < W;
        pix = ((long long)pb * H + py) * W + px;
      }
      float b[COT];
#pragma unroll
      for (int ct = 0; ct < COT; ++ct) b[ct] = (ok && co0 + ct * 16 < cout) ? gz[pix * cout + co0 + ct * 16 + (l & 15)] : 0.0f;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const float a = lds[T.lds_pixel(q, t) * CKP + (l & 15)];
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[ct], acc[t][ct], 0, 0, 0);
      }
    }
  }
  // the four waves' sums, added in wave order through LDS, then one partial per workgroup
  float* out = partial + (long long)blockIdx.x * TAPS * cin16 * cout;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
      __syncthreads();
      *(floatx4*)&lds[(wv * 64 + l) * 4] = acc[t][ct];
      __syncthreads();
      if (wv == 0) {
        floatx4 s = *(floatx4*)&lds[l * 4];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const floatx4 o = *(floatx4*)&lds[(w * 64 + l) * 4]; s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3]; }
        const int co = co0 + ct * 16 + (l & 15);
        if (co < cout)
#pragma unroll
          for (int i = 0; i < 4; ++i) out[((long long)t * cin16 + ci0 + 4 * (l >> 4) + i) * cout + co] = s[i];
      }
    }
}

// gw (torch layout) = sum over the G partials in index order.  kind 0: Conv2d weight [cout][cin][3][3] (or 1 x 1);
// kind 1: ConvTranspose2d weight [cin][cout_t][2][2] from the 1 x 1 form with cout = 4 cout_t channels k = q cout_t + co.
// A workgroup takes 32 consecutive elements; its 256 threads are (row r of 8, element): row r adds the partials
// r, r + 8, ... in turn (128-byte reads), then the rows are added in index order — a fixed summation order.
// Blocks past the weight elements (nwb of them) finish the layer's bias gradient from the partials k_tact_bwd left
// (bpart [bnblk][bC], 32 channels per block): one launch per layer instead of two.
__global__ void __launch_bounds__(256) k_twrw_finish(const float* __restrict__ partial, int G, int taps, int cin, int cout,
                                                     int kind, float* __restrict__ gw, int nwb, const float* __restrict__ bpart,
                                                     int bnblk, int bC, float* __restrict__ gb) {
  __shared__ float sh[256];
  if ((int)blockIdx.x >= nwb) {
    const int c = ((int)blockIdx.x - nwb) * 32 + (threadIdx.x & 31), r = threadIdx.x >> 5;
    float s = 0.0f;
    if (c < bC)
      for (int b = r; b < bnblk; b += 8) s += bpart[(long long)b * bC + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 32 && c < bC) {
      float t8 = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t8 += sh[q * 32 + threadIdx.x];
      gb[c] = t8;
    }
    return;
  }
  const int cin16 = (cin + 15) & ~15;
  const long long n = (long long)taps * cin16 * cout, e = (long long)blockIdx.x * 32 + (threadIdx.x & 31);
  const int r = threadIdx.x >> 5;
  float s = 0.0f;
  if (e < n)
    for (int g = r; g < G; g += 8) s += partial[(long long)g * n + e];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x >= 32 || e >= n) return;
  float t8 = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) t8 += sh[q * 32 + threadIdx.x];
  const int co = (int)(e % cout), ci = (int)((e / cout) % cin16), t = (int)(e / ((long long)cout * cin16));
  if (ci >= cin) return;
  if (kind == 0) gw[((long long)co * cin + ci) * taps + t] = t8;
  else { const int ct = cout / 4, q = co / ct, c = co - q * ct; gw[((long long)ci * ct + c) * 4 + q] = t8; }
}

// ---------------------------------------------------------------------------------------------------------------- act bwd
// gz[p][c] = (g[p][c] + routed pool gradient) * [y[p][c] > 0]; thread = (pixel, 4-channel group).  gp (may be NULL):
// gradient of the 2 x 2 max-pooled map [B][H/2][W/2][C]: it goes to the first maximal element of the window in row-major
// order (the library's rule).  s2d: gz is stored space-to-depth, [B][H/2][W/2][4 C] with channel q C + c, q = 2 (y % 2) +
// x % 2.  Bias-gradient partials per block (fixed order), finished by k_tbias_finish.
__global__ void __launch_bounds__(256) k_tact_bwd(const float* __restrict__ g, int gs, int go, const float* __restrict__ y,
                                                  int ys, int yo, const float* __restrict__ gp, float* __restrict__ gz,
                                                  float* __restrict__ partial, int B, int H, int W, int C, int relu,
                                                  int s2d, int pixb) {
  __shared__ float sh[256 * 4];
  const int cg = C / 4, ppi = 256 / cg;
  const int gi = threadIdx.x % cg, pl = threadIdx.x / cg;
  const long long npix = (long long)B * H * W;
  const long long p0 = (long long)blockIdx.x * pixb;
  long long p1 = p0 + pixb; if (p1 > npix) p1 = npix;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (pl < ppi)
    for (long long p = p0 + pl; p < p1; p += ppi) {
      const float4 gv = *(const float4*)(g + p * gs + go + gi * 4);
      float a[4] = {gv.x, gv.y, gv.z, gv.w};
      const int xx = (int)(p % W); const long long r = p / W; const int yy = (int)(r % H), b = (int)(r / H);
      float4 yv = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
      if (relu || gp) yv = *(const float4*)(y + p * ys + yo + gi * 4);
      if (gp) {
        const int wy = yy & ~1, wx = xx & ~1, me = 2 * (yy & 1) + (xx & 1);
        const float4 pv = *(const float4*)(gp + ((((long long)b * (H / 2) + (yy >> 1)) * (W / 2)) + (xx >> 1)) * C + gi * 4);
        float w4[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = *(const float4*)(y + ((((long long)b * H + wy + (q >> 1)) * W) + wx + (q & 1)) * ys + yo + gi * 4);
          w4[q][0] = t.x; w4[q][1] = t.y; w4[q][2] = t.z; w4[q][3] = t.w;
        }
        const float pg[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int best = 0; float bv = w4[0][k];
#pragma unroll
          for (int q = 1; q < 4; ++q) if (w4[q][k] > bv) { bv = w4[q][k]; best = q; }
          if (best == me) a[k] += pg[k];
        }
      }
      if (relu) { a[0] = yv.x > 0.0f ? a[0] : 0.0f; a[1] = yv.y > 0.0f ? a[1] : 0.0f; a[2] = yv.z > 0.0f ? a[2] : 0.0f; a[3] = yv.w > 0.0f ? a[3] : 0.0f; }
      long long o = p * C + gi * 4;
      if (s2d) o = ((((long long)b * (H / 2) + (yy >> 1)) * (W / 2)) + (xx >> 1)) * 4 * C + (2 * (yy & 1) + (xx & 1)) * C + gi * 4;
      *(float4*)(gz + o) = make_float4(a[0], a[1], a[2], a[3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += a[k];
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) sh[threadIdx.x * 4 + k] = acc[k];
  __syncthreads();
  if (partial && threadIdx.x < C) {
    const int c = threadIdx.x, gg = c / 4, k = c % 4;
    float s = 0.0f;
    for (int q = 0; q < ppi; ++q) s += sh[(q * cg + gg) * 4 + k];
    partial[(long long)blockIdx.x * C + c] = s;
  }
}

__global__ void __launch_bounds__(256) k_tbias_finish(const float* __restrict__ partial, int nblk, int C, float* __restrict__ gb) {
  __shared__ float sh[256];
  const int cw = C < 32 ? C : 32, rows = 256 / cw;
  const int c = blockIdx.x * cw + threadIdx.x % cw, r = threadIdx.x / cw;
  float s = 0.0f;
  if (r < rows && c < C)
    for (int b = r; b < nblk; b += rows) s += partial[(long long)b * C + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < cw && c < C) {
    float t = 0.0f;
    for (int q = 0; q < rows; ++q) t += sh[q * cw + threadIdx.x];
    gb[c] = t;
  }
}

int act_pixb(long long npix, int C) {
  const int ppi = 256 / (C / 4);
  long long pixb = (npix + 1023) / 1024;
  if (pixb < 4 * ppi) pixb = 4 * ppi;
  return (int)((pixb + ppi - 1) / ppi * ppi);
}

// ---------------------------------------------------------------------------------------------------------------- repack
// desc[i] = {src_off, dst_off, cin, cout, taps, kind, cin_pad (kind 1; 0 = cin), 0} (int64 x 8); one thread per destination element.
//   kind 0  conv forward     dst[(t cin_p + ci) cout + co]       = w[co][ci][t]            (ci >= cin: 0), cin_p = cin up to 4
//   kind 1  conv data grad   dst[(t cout + co) cin_pad + ci]     = w[co][ci][taps - 1 - t]   (ci >= cin: 0)
//   kind 2  convT forward    dst[ci 4 cout + q cout + co]        = w[ci][co][q]            (ConvTranspose2d [cin][cout][2][2])
//   kind 3  convT data grad  dst[(q cout + co) cin + ci]         = w[ci][co][q]
__global__ void __launch_bounds__(256) k_trepack(const float* __restrict__ flat, float* __restrict__ packed,
                                                 const long long* __restrict__ desc, int nlayers, long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  int lo = 0, hi = nlayers - 1;       // the layer whose destination range holds e (ranges are consecutive)
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid * 8 + 1] <= e) lo = mid; else hi = mid - 1; }
  const long long* d = desc + lo * 8;
  const long long src = d[0], r = e - d[1];
  const int cin = (int)d[2], cout = (int)d[3], taps = (int)d[4], kind = (int)d[5];
  float v = 0.0f;
  if (kind == 0) {
    const int cin_p = (cin + 3) & ~3;
    const int co = (int)(r % cout), ci = (int)((r / cout) % cin_p), t = (int)(r / ((long long)cout * cin_p));
    if (ci < cin) v = flat[src + ((long long)co * cin + ci) * taps + t];
  } else if (kind == 1) {
    const int cinp = d[6] ? (int)d[6] : cin;      // the data gradient's output channels, padded to a multiple of 16 (zeros)
    const int ci = (int)(r % cinp), co = (int)((r / cinp) % cout), t = (int)(r / ((long long)cinp * cout));
    if (ci < cin) v = flat[src + ((long long)co * cin + ci) * taps + (taps - 1 - t)];
  } else if (kind == 2) {
    const int k = (int)(r % (4 * cout)), ci = (int)(r / (4 * cout)), q = k / cout, co = k - q * cout;
    v = flat[src + ((long long)ci * cout + co) * 4 + q];
  } else {
    const int ci = (int)(r % cin), k = (int)(r / cin), q = k / cout, co = k - q * cout;
    v = flat[src + ((long long)ci * cout + co) * 4 + q];
  }
  packed[e] = v;
}

template <int TAPS, int TW>
int launch_conv(const float* x, int xs, int xo, const float* wp, const float* bias, float* y, int ys, int yo, int B, int H,
                int W, int cin, int cout, int relu, int d2s, hipStream_t st) {
  long long tiles;
  if (TAPS == 1) tiles = ((long long)B * H * W + 255) / 256;
  else if (TW == 16) tiles = (long long)B * ((H + 15) / 16) * ((W + 15) / 16);
  else tiles = (long long)((B + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8);
  // output channels per workgroup: 64 when that still gives the chip >= 512 workgroups, else 32, else 16 (the deep levels
  // have few pixel tiles: a 256 -> 256 layer at 8 x 8 x 64 samples is 16 tiles)
  if (cout % 64 == 0 && tiles * (cout / 64) >= 512)
    hipLaunchKernelGGL((k_tconv<TAPS, TW, 4>), dim3((unsigned)tiles, cout / 64), dim3(256), 0, st, x, xs, xo, wp, bias, y, ys, yo, B, H, W, cin, cout, relu, d2s);
  else if (cout % 32 == 0 && tiles * (cout / 32) >= 512)
    hipLaunchKernelGGL((k_tconv<TAPS, TW, 2>), dim3((unsigned)tiles, cout / 32), dim3(256), 0, st, x, xs, xo, wp, bias, y, ys, yo, B, H, W, cin, cout, relu, d2s);
  else
    hipLaunchKernelGGL((k_tconv<TAPS, TW, 1>), dim3((unsigned)tiles, cout / 16), dim3(256), 0, st, x, xs, xo, wp, bias, y, ys, yo, B, H, W, cin, cout, relu, d2s);
  return t_finish("srl_tconv");
}

long long wrw_tiles(int taps, int B, int H, int W) {
  if (taps == 1) return ((long long)B * H * W + 255) / 256;
  if (W > 8 || H > 8) return (long long)B * ((H + 15) / 16) * ((W + 15) / 16);
  return (long long)((B + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8);
}

// pixel-tile groups: about 1,024 workgroups per launch, at most one group per tile
int wrw_groups(int taps, int B, int H, int W, int cin, int cout) {
  const long long tiles = wrw_tiles(taps, B, H, W);
  const int cot = cout % 64 == 0 ? 4 : cout % 32 == 0 ? 2 : 1;
  const long long blocks = (long long)((cin + 15) / 16) * (cout / (cot * 16));
  long long G = 1024 / blocks;
  if (G < 1) G = 1;
  if (G > tiles) G = tiles;
  if (G > 512) G = 512;
  return (int)G;
}

template <int TAPS, int TW>
int launch_wrw(const float* x, int xs, int xo, const float* gz, float* partial, int G, int B, int H, int W, int cin, int cout,
               hipStream_t st) {
  const long long tiles = wrw_tiles(TAPS, B, H, W);
  const int nci = (cin + 15) / 16;
  if (cout % 64 == 0)
    hipLaunchKernelGGL((k_twrw<TAPS, TW, 4>), dim3(G, nci, cout / 64), dim3(256), 0, st, x, xs, xo, gz, partial, B, H, W, cin, cout, tiles);
  else if (cout % 32 == 0)
    hipLaunchKernelGGL((k_twrw<TAPS, TW, 2>), dim3(G, nci, cout / 32), dim3(256), 0, st, x, xs, xo, gz, partial, B, H, W, cin, cout, tiles);
  else
    hipLaunchKernelGGL((k_twrw<TAPS, TW, 1>), dim3(G, nci, cout / 16), dim3(256), 0, st, x, xs, xo, gz, partial, B, H, W, cin, cout, tiles);
  return t_finish("srl_twrw");
}


// ---------------------------------------------------------------------------------------------------------------- dueling head
// Q(s, .) of `DeepQSiamFCN` from the last position map z [B][A][16] (channels last, A = O x O pixels), the 1 x 1 projection
// (weight pw[16], bias pb) and the state value v[B] (models.py:179-192): a[b][p] = z[b][p][:] . pw + pb,
// q[b][p] = a[b][p] - mean_p a[b][.] + v[b].  One workgroup per sample, every sum in a fixed order (per-thread strided
// partial sums, then a halving tree over the 256 partials): no cross-workgroup reduction, no atomics — the framework's
// multi-block reductions are what returned garbage under the concurrent env step.
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256) k_thead_fwd(const float* __restrict__ z, const float* __restrict__ pw,
                                                   const float* __restrict__ pb, const float* __restrict__ v,
                                                   float* __restrict__ q, int A) {
  __shared__ float sh[256];
  const int b = blockIdx.x;
  const float4* zb = (const float4*)(z + (long long)b * A * 16);
  float* qb = q + (long long)b * A;
  float w[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) w[c] = pw[c];
  const float bias = pb[0];
  float s = 0.0f;
  for (int p = threadIdx.x; p < A; p += 256) {
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 t = zb[4 * p + k];
      a = fmaf(t.x, w[4 * k], a); a = fmaf(t.y, w[4 * k + 1], a); a = fmaf(t.z, w[4 * k + 2], a); a = fmaf(t.w, w[4 * k + 3], a);
    }
    a += bias;
    qb[p] = a;
    s += a;
  }
  const float mean = block_sum_256(s, sh) / (float)A;
  const float add = v[b] - mean;
  for (int p = threadIdx.x; p < A; p += 256) qb[p] += add;
}

// backward of the above for the first n samples: gq[n][A] -> gv[b] = sum_p gq[b][p], ga = gq - mean_p gq,
// gz[b][p][c] = ga pw[c], and per-sample partials part[b][0..15] = sum_p z[b][p][c] ga, part[b][16] = sum_p ga
__global__ void __launch_bounds__(256) k_thead_bwd(const float* __restrict__ z, const float* __restrict__ pw,
                                                   const float* __restrict__ gq, float* __restrict__ gz,
                                                   float* __restrict__ gv, float* __restrict__ part, int A) {
  __shared__ float sh[256];
  const int b = blockIdx.x;
  const float4* zb = (const float4*)(z + (long long)b * A * 16);
  float4* gzb = (float4*)(gz + (long long)b * A * 16);
  const float* gb = gq + (long long)b * A;
  float w[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) w[c] = pw[c];
  float s = 0.0f;
  for (int p = threadIdx.x; p < A; p += 256) s += gb[p];
  const float tot = block_sum_256(s, sh);
  if (threadIdx.x == 0) gv[b] = tot;
  const float mean = tot / (float)A;
  float acc[17];
#pragma unroll
  for (int c = 0; c < 17; ++c) acc[c] = 0.0f;
  for (int p = threadIdx.x; p < A; p += 256) {
    const float ga = gb[p] - mean;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 t = zb[4 * p + k];
      acc[4 * k] = fmaf(t.x, ga, acc[4 * k]); acc[4 * k + 1] = fmaf(t.y, ga, acc[4 * k + 1]);
      acc[4 * k + 2] = fmaf(t.z, ga, acc[4 * k + 2]); acc[4 * k + 3] = fmaf(t.w, ga, acc[4 * k + 3]);
      gzb[4 * p + k] = make_float4(ga * w[4 * k], ga * w[4 * k + 1], ga * w[4 * k + 2], ga * w[4 * k + 3]);
    }
    acc[16] += ga;
  }
#pragma unroll
  for (int c = 0; c < 17; ++c) {
    const float r = block_sum_256(acc[c], sh);
    if (threadIdx.x == 0) part[b * 17 + c] = r;
  }
}

// gpw[c] = sum_b part[b][c], gpb = sum_b part[b][16], samples in index order
__global__ void __launch_bounds__(64) k_thead_finish(const float* __restrict__ part, int n, float* __restrict__ gpw,
                                                     float* __restrict__ gpb) {
  const int c = threadIdx.x;
  if (c > 16) return;
  float s = 0.0f;
  for (int b = 0; b < n; ++b) s += part[b * 17 + c];
  if (c < 16) gpw[c] = s; else gpb[0] = s;
}

// ---------------------------------------------------------------------------------------------------------------- value head
// `layers.value` (layers.py:424-436) of the dueling network (models.py:179-192): global average pool of the bottom features
// x0 [B][P][C] (P = h x w pixels) -> Dense(U) + ReLU -> Dense(1).  Rounds 1-3 left these two small dense layers to the
// framework (rocBLAS GEMMs, bias / ReLU / reduction kernels through autograd).  Here: one workgroup per sample forwards,
// one per sample + one per hidden unit backwards; every sum runs in index order (pixels, inputs, samples), no atomics.
//   pooled[b][i] = (sum_p x0[b][p][i]) / P;  h[b][j] = relu(b1[j] + sum_i pooled[b][i] W1[j][i]);  v[b] = b2 + sum_j h[b][j] W2[j]
__global__ void __launch_bounds__(256) k_tvalue_fwd(const float* __restrict__ x0, const float* __restrict__ W1,
                                                    const float* __restrict__ b1, const float* __restrict__ W2,
                                                    const float* __restrict__ b2, float* __restrict__ pooled,
                                                    float* __restrict__ h, float* __restrict__ v, int P, int C, int U) {
  extern __shared__ float sh[];                      // [C] pooled features, then [256] partial sums
  float* red = sh + C;
  const int b = blockIdx.x;
  const float* xb = x0 + (long long)b * P * C;
  for (int i = threadIdx.x; i < C; i += 256) {
    float s = 0.0f;
    for (int p = 0; p < P; ++p) s += xb[(long long)p * C + i];
    s = s / (float)P;
    sh[i] = s;
    if (pooled) pooled[(long long)b * C + i] = s;
  }
  __syncthreads();
  float part = 0.0f;
  for (int j = threadIdx.x; j < U; j += 256) {
    const float* w = W1 + (long long)j * C;
    float a = b1[j];
    for (int i = 0; i < C; ++i) a = fmaf(sh[i], w[i], a);
    a = fmaxf(a, 0.0f);
    if (h) h[(long long)b * U + j] = a;
    part = fmaf(a, W2[j], part);
  }
  const float tot = block_sum_256(part, red);
  if (threadIdx.x == 0) v[b] = tot + b2[0];
}

// backward for the first n samples, kernel A (one workgroup per sample): gh[b][j] = gv[b] W2[j] [h[b][j] > 0] (kept for kernel
// B), gpooled[i] = sum_j gh[j] W1[j][i], and the gradient wrt x0: gx[b][p][i] = gin[b][p][i] + gpooled[i] / P (gin = the
// gradient that reaches x0 through the decoder; the two meet here instead of in a framework add)
__global__ void __launch_bounds__(256) k_tvalue_bwd_a(const float* __restrict__ gv, const float* __restrict__ h,
                                                      const float* __restrict__ W1, const float* __restrict__ W2,
                                                      const float* __restrict__ gin, float* __restrict__ gh,
                                                      float* __restrict__ gx, int P, int C, int U) {
  extern __shared__ float sh[];                      // [U] gh of this sample
  const int b = blockIdx.x;
  const float g = gv[b];
  for (int j = threadIdx.x; j < U; j += 256) {
    const float t = h[(long long)b * U + j] > 0.0f ? g * W2[j] : 0.0f;
    sh[j] = t;
    gh[(long long)b * U + j] = t;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    float a = 0.0f;
    for (int j = 0; j < U; ++j) a = fmaf(sh[j], W1[(long long)j * C + i], a);
    a = a / (float)P;
    for (int p = 0; p < P; ++p) {
      const long long o = ((long long)b * P + p) * C + i;
      gx[o] = gin[o] + a;
    }
  }
}

// kernel B: block j < U: gW1[j][i] = sum_b gh[b][j] pooled[b][i] (thread i), gb1[j] = sum_b gh[b][j]; block U: gW2[j] =
// sum_b gv[b] h[b][j] (thread j), gb2 = sum_b gv[b].  Samples in index order.
__global__ void __launch_bounds__(256) k_tvalue_bwd_b(const float* __restrict__ gv, const float* __restrict__ gh,
                                                      const float* __restrict__ h, const float* __restrict__ pooled,
                                                      float* __restrict__ gW1, float* __restrict__ gb1,
                                                      float* __restrict__ gW2, float* __restrict__ gb2, int n, int C, int U) {
  const int j = blockIdx.x;
  if (j < U) {
    for (int i = threadIdx.x; i < C; i += 256) {
      float a = 0.0f;
      for (int b = 0; b < n; ++b) a = fmaf(gh[(long long)b * U + j], pooled[(long long)b * C + i], a);
      gW1[(long long)j * C + i] = a;
    }
    if (threadIdx.x == 0) {
      float a = 0.0f;
      for (int b = 0; b < n; ++b) a += gh[(long long)b * U + j];
      gb1[j] = a;
    }
    return;
  }
  for (int k = threadIdx.x; k < U; k += 256) {
    float a = 0.0f;
    for (int b = 0; b < n; ++b) a = fmaf(gv[b], h[(long long)b * U + k], a);
    gW2[k] = a;
  }
  if (threadIdx.x == 0) {
    float a = 0.0f;
    for (int b = 0; b < n; ++b) a += gv[b];
    gb2[0] = a;
  }
}

// ---------------------------------------------------------------------------------------------------------------- layout passes
// The cross-correlation kernels (csrc/xcorr_mfma.hip) read channel-major maps [B][C][H][W]; the U-Nets of the update work on
// channels-last ones [B][H][W][C].  These were framework copies (permute + contiguous, pad, flip); one kernel each:
//   k_tlayout mode 0: NHWC (stride / offset slice) -> NCHW;  mode 1: NCHW -> NHWC;  both one element per thread, the reads
//   of a wave contiguous in the source's fastest dimension for mode 1, in the destination's for mode 0
__global__ void __launch_bounds__(256) k_tlayout(const float* __restrict__ src, int ss, int so, float* __restrict__ dst, int B,
                                                 int HW, int C, int mode) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x, n = (long long)B * HW * C;
  if (e >= n) return;
  if (mode == 0) {          // e indexes dst [b][c][p]
    const int p = (int)(e % HW); const long long r = e / HW; const int c = (int)(r % C), b = (int)(r / C);
    dst[e] = src[((long long)b * HW + p) * ss + so + c];
  } else {                  // e indexes dst [b][p][c]; src [b][c][p]
    const int c = (int)(e % C); const long long r = e / C; const int p = (int)(r % HW), b = (int)(r / HW);
    dst[e] = src[((long long)b * C + c) * HW + p];
  }
}

// channel 0 of a channels-last gradient g [n][O][O][Cs] -> the plain map [n][O][O] AND its zero-padded copy
// [n][O + 2 pad][O + 2 pad] (the operand of the cross-correlation's data gradient)
__global__ void __launch_bounds__(256) k_tcorr_grad(const float* __restrict__ g, int Cs, float* __restrict__ plain,
                                                    float* __restrict__ padded, int n, int O, int pad) {
  const int Op = O + 2 * pad;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x, tot = (long long)n * Op * Op;
  if (e >= tot) return;
  const int x = (int)(e % Op) - pad; const long long r = e / Op; const int y = (int)(r % Op) - pad, b = (int)(r / Op);
  float v = 0.0f;
  if (x >= 0 && x < O && y >= 0 && y < O) {
    v = g[(((long long)b * O + y) * O + x) * Cs];
    plain[((long long)b * O + y) * O + x] = v;
  }
  padded[e] = v;
}

// out[b][c][kh - 1 - y][kw - 1 - x] = in[b][c][y][x] for the first n samples (the flipped kernels of the data gradient)
__global__ void __launch_bounds__(256) k_tflip(const float* __restrict__ in, float* __restrict__ out, long long planes, int hw) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= planes * hw) return;
  const long long pl = e / hw; const int k = (int)(e % hw);
  out[pl * hw + (hw - 1 - k)] = in[e];
}

// uint8 observation -> float32 / 255 (models.py:144-147), 4 elements per thread
__global__ void __launch_bounds__(256) k_tu8_to_f32(const uint8_t* __restrict__ in, float* __restrict__ out, long long n4) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  const uchar4 v = ((const uchar4*)in)[e];
  ((float4*)out)[e] = make_float4((float)v.x / 255.0f, (float)v.y / 255.0f, (float)v.z / 255.0f, (float)v.w / 255.0f);
}

}  // namespace

extern "C" {

const char* srl_train_conv_last_error(void) { return t_err; }

int srl_tconv(const float* x, int32_t x_stride, int32_t x_off, const float* wp, const float* bias, float* y, int32_t y_stride,
              int32_t y_off, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, int32_t relu,
              int32_t d2s, void* stream) {
  if (!x || !wp || !y || B < 1 || H < 1 || W < 1 || cin < 1 || cout < 16 || cout % 16 || (taps != 1 && taps != 9) ||
      y_stride % 4 || y_off % 4 || (d2s && (taps != 1 || cout != 4 * d2s || d2s % 4)))
    return t_bad("srl_tconv: bad arguments (cout a multiple of 16, taps 1 or 9, output stride / offset multiples of 4)");
  hipStream_t st = (hipStream_t)stream;
  if (taps == 1) return launch_conv<1, 16>(x, x_stride, x_off, wp, bias, y, y_stride, y_off, B, H, W, cin, cout, relu, d2s, st);
  if (W > 8 || H > 8) return launch_conv<9, 16>(x, x_stride, x_off, wp, bias, y, y_stride, y_off, B, H, W, cin, cout, relu, 0, st);
  return launch_conv<9, 8>(x, x_stride, x_off, wp, bias, y, y_stride, y_off, B, H, W, cin, cout, relu, 0, st);
}

int64_t srl_twrw_scratch_floats(int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps) {
  if (B < 1 || H < 1 || W < 1 || cin < 1 || cout < 16 || cout % 16 || (taps != 1 && taps != 9)) return -1;
  return (int64_t)wrw_groups(taps, B, H, W, cin, cout) * taps * ((cin + 15) & ~15) * cout;
}

int srl_twrw(const float* x, int32_t x_stride, int32_t x_off, const float* gz, float* gw, float* scratch, int32_t B,
             int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, int32_t convt, const float* bias_partial,
             int32_t bias_nblk, int32_t bias_C, float* gbias, void* stream) {
  if (!x || !gz || !gw || !scratch || srl_twrw_scratch_floats(B, H, W, cin, cout, taps) < 0 || (convt && (taps != 1 || cout % 4)))
    return t_bad("srl_twrw: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int G = wrw_groups(taps, B, H, W, cin, cout);
  int rc;
  if (taps == 1) rc = launch_wrw<1, 16>(x, x_stride, x_off, gz, scratch, G, B, H, W, cin, cout, st);
  else if (W > 8 || H > 8) rc = launch_wrw<9, 16>(x, x_stride, x_off, gz, scratch, G, B, H, W, cin, cout, st);
  else rc = launch_wrw<9, 8>(x, x_stride, x_off, gz, scratch, G, B, H, W, cin, cout, st);
  if (rc) return rc;
  const long long n = (long long)taps * ((cin + 15) & ~15) * cout;
  const int nwb = (int)((n + 31) / 32), nbb = (bias_partial && gbias) ? (bias_C + 31) / 32 : 0;
  hipLaunchKernelGGL(k_twrw_finish, dim3(nwb + nbb), dim3(256), 0, st, scratch, G, taps, cin, cout, convt ? 1 : 0, gw, nwb,
                     bias_partial, bias_nblk, bias_C, gbias);
  return t_finish("srl_twrw finish");
}

int64_t srl_tact_bwd_scratch_floats(int64_t npix, int32_t C) {
  if (npix < 1 || C < 4 || C % 4 || C > 256 || 256 % (C / 4)) return -1;
  const int pixb = act_pixb(npix, C);
  return ((npix + pixb - 1) / pixb) * C;
}

int32_t srl_tact_bwd_blocks(int64_t npix, int32_t C) {
  if (srl_tact_bwd_scratch_floats(npix, C) < 0) return -1;
  const int pixb = act_pixb(npix, C);
  return (int32_t)((npix + pixb - 1) / pixb);
}

int srl_tact_bwd(const float* g, int32_t g_stride, int32_t g_off, const float* y, int32_t y_stride, int32_t y_off,
                 const float* gpool, float* gz, float* gbias, float* scratch, int32_t B, int32_t H, int32_t W, int32_t C,
                 int32_t relu, int32_t s2d, void* stream) {
  const long long npix = (long long)B * H * W;
  if (!g || !gz || (!y && (relu || gpool)) || srl_tact_bwd_scratch_floats(npix, C) < 0 || g_stride % 4 || g_off % 4 ||
      y_stride % 4 || y_off % 4 || ((gpool || s2d) && ((H | W) & 1)))
    return t_bad("srl_tact_bwd: bad arguments (C a multiple of 4 with C / 4 dividing 256, strides / offsets multiples of 4)");
  const int pixb = act_pixb(npix, C);
  const int nblk = (int)((npix + pixb - 1) / pixb);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_tact_bwd, dim3(nblk), dim3(256), 0, st, g, g_stride, g_off, y, y_stride, y_off, gpool, gz,
                     scratch, B, H, W, C, relu, s2d, pixb);
  if (gbias) hipLaunchKernelGGL(k_tbias_finish, dim3((C + 31) / 32), dim3(256), 0, st, scratch, nblk, C, gbias);
  return t_finish("srl_tact_bwd");
}

int srl_trepack(const float* flat, float* packed, const int64_t* desc_dev, int32_t nlayers, int64_t total, void* stream) {
  if (!flat || !packed || !desc_dev || nlayers < 1 || total < 1) return t_bad("srl_trepack: bad arguments");
  hipLaunchKernelGGL(k_trepack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat, packed,
                     (const long long*)desc_dev, nlayers, (long long)total);
  return t_finish("srl_trepack");
}

int srl_thead_fwd(const float* z, const float* pw, const float* pb, const float* v, float* q, int32_t B, int32_t A, void* stream) {
  if (!z || !pw || !pb || !v || !q || B < 1 || A < 1) return t_bad("srl_thead_fwd: bad arguments");
  hipLaunchKernelGGL(k_thead_fwd, dim3(B), dim3(256), 0, (hipStream_t)stream, z, pw, pb, v, q, A);
  return t_finish("srl_thead_fwd");
}

int srl_thead_bwd(const float* z, const float* pw, const float* gq, float* gz, float* gv, float* gpw, float* gpb,
                  float* scratch, int32_t n, int32_t A, void* stream) {
  if (!z || !pw || !gq || !gz || !gv || !gpw || !gpb || !scratch || n < 1 || A < 1) return t_bad("srl_thead_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_thead_bwd, dim3(n), dim3(256), 0, st, z, pw, gq, gz, gv, scratch, A);
  hipLaunchKernelGGL(k_thead_finish, dim3(1), dim3(64), 0, st, scratch, n, gpw, gpb);
  return t_finish("srl_thead_bwd");
}

int srl_tvalue_fwd(const float* x0, const float* W1, const float* b1, const float* W2, const float* b2, float* pooled, float* h,
                   float* v, int32_t B, int32_t P, int32_t C, int32_t U, void* stream) {
  if (!x0 || !W1 || !b1 || !W2 || !b2 || !v || B < 1 || P < 1 || C < 1 || U < 1 || C > 4096) return t_bad("srl_tvalue_fwd: bad arguments");
  hipLaunchKernelGGL(k_tvalue_fwd, dim3(B), dim3(256), sizeof(float) * (C + 256), (hipStream_t)stream, x0, W1, b1, W2, b2, pooled, h, v, P, C, U);
  return t_finish("srl_tvalue_fwd");
}

int srl_tvalue_bwd(const float* gv, const float* h, const float* pooled, const float* W1, const float* W2, const float* gin,
                   float* gx, float* gW1, float* gb1, float* gW2, float* gb2, float* scratch, int32_t n, int32_t P, int32_t C,
                   int32_t U, void* stream) {
  if (!gv || !h || !pooled || !W1 || !W2 || !gin || !gx || !gW1 || !gb1 || !gW2 || !gb2 || !scratch || n < 1 || P < 1 || C < 1 ||
      U < 1 || U > 4096)
    return t_bad("srl_tvalue_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_tvalue_bwd_a, dim3(n), dim3(256), sizeof(float) * U, st, gv, h, W1, W2, gin, scratch, gx, P, C, U);
  hipLaunchKernelGGL(k_tvalue_bwd_b, dim3(U + 1), dim3(256), 0, st, gv, (const float*)scratch, h, pooled, gW1, gb1, gW2, gb2, n, C, U);
  return t_finish("srl_tvalue_bwd");
}

int srl_tlayout(const float* src, int32_t src_stride, int32_t src_off, float* dst, int32_t B, int32_t HW, int32_t C, int32_t to_nhwc,
                void* stream) {
  if (!src || !dst || B < 1 || HW < 1 || C < 1) return t_bad("srl_tlayout: bad arguments");
  const long long n = (long long)B * HW * C;
  hipLaunchKernelGGL(k_tlayout, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, src_stride, src_off, dst, B, HW, C,
                     to_nhwc ? 1 : 0);
  return t_finish("srl_tlayout");
}

int srl_tcorr_grad(const float* g, int32_t g_stride, float* plain, float* padded, int32_t n, int32_t O, int32_t pad, void* stream) {
  if (!g || !plain || !padded || n < 1 || O < 1 || pad < 0 || g_stride < 1) return t_bad("srl_tcorr_grad: bad arguments");
  const long long tot = (long long)n * (O + 2 * pad) * (O + 2 * pad);
  hipLaunchKernelGGL(k_tcorr_grad, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, g_stride, plain, padded, n, O, pad);
  return t_finish("srl_tcorr_grad");
}

int srl_tflip(const float* in, float* out, int64_t planes, int32_t hw, void* stream) {
  if (!in || !out || planes < 1 || hw < 1) return t_bad("srl_tflip: bad arguments");
  hipLaunchKernelGGL(k_tflip, dim3((unsigned)((planes * hw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, (long long)planes, hw);
  return t_finish("srl_tflip");
}

int srl_tu8_to_f32(const uint8_t* in, float* out, int64_t n, void* stream) {
  if (!in || !out || n < 4 || n % 4) return t_bad("srl_tu8_to_f32: bad arguments (element count a multiple of 4)");
  hipLaunchKernelGGL(k_tu8_to_f32, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, (long long)(n / 4));
  return t_finish("srl_tu8_to_f32");
}

}  // extern "C"
