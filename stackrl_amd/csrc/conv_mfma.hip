// conv_mfma.hip — 3 x 3 convolution (stride 1, SAME) + bias + ReLU of the U-Nets' wide, thin layers on the matrix
// cores, inference only, gfx950.  Replaces Conv2D(f, 3, padding='same', activation='relu') [+ MaxPool2D / Concatenate]
// of `layers.unet` (stackrl/nets/layers.py:135-259) for the 16- and 32-channel layers at 128^2 .. 16^2, which carry
// most of the forward's activation traffic and are bound by HBM, not by MFMA: the library route writes the raw
// convolution output and re-reads it for bias / ReLU / pooling; here one kernel reads the input once and writes the
// finished tensor(s) once.
//
// Implicit GEMM, D[cout][pixel] += W[cout][k] X[k][pixel], k = (tap, cin), v_mfma_f32_16x16x32_bf16 (K = 32 per
// instruction: one tap x 32 channels, or two taps x 16 channels).  bf16 channels-last in and out, fp32 accumulation.
//   workgroup = 4 waves = a 16 x 16 pixel tile of one image; wave = 4 rows of 16 pixels x all output channels
//   input tile + halo (18 x 18 pixels x CIN) staged once in LDS (zero padding at the image border); a B fragment is
//   the 8 consecutive channels of one pixel (16 bytes, ds_read_b128)
//   weights: pre-packed in A-fragment order (host side, once per weight update) and held in registers for the whole
//   kernel (<= 72 VGPRs for 32 -> 32)
//   epilogue: bias + ReLU on the 4 consecutive output channels a lane holds, 8-byte stores into a channel slice of a
//   channels-last buffer (the decoder's concat buffer), optionally the 2 x 2 max-pooled tensor too (rows in registers,
//   columns by a DPP quad permute), or channel-major output for the cross-correlation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stackrl_qnet.h"
#include "srl_bf16.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The four lane groups of a wave hold the four channel quarters of one pixel: (q0 + q1) + (q2 + q3) into lane group 0 (the
// other groups' results are not used), the partners' values by v_permlane16_swap / v_permlane32_swap instead of two
// ds_bpermute round trips (`__shfl_xor` 16, 32) — the same sums in the same order.
__device__ __forceinline__ float quarter_sum(float v) {
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = v + __uint_as_float(r[1]);   // rows 0 and 2 take rows 1 and 3
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = v + __uint_as_float(r[1]);   // lanes 0..31 take lanes 32..63
  }
  return v;
}

template <int CIN>
struct ConvCfg {
  static constexpr int KS = CIN == 16 ? 5 : 9 * (CIN / 32);   // K = 32 steps: tap pairs (the 10th tap is zero) | taps x 32-channel blocks
  static constexpr int PS = CIN + 8;                          // LDS pixel stride in elements (16-byte aligned, de-phased banks)
  static constexpr int TW = 18;                               // tile width / height incl. halo
};

// tap and first channel of the 8 k-values lane group g (0..3) holds in K-step ks
template <int CIN>
__device__ __forceinline__ void k_of(int ks, int g, int& tap, int& ci0) {
  if (CIN == 16) { tap = 2 * ks + (g >> 1); ci0 = 8 * (g & 1); }
  else { tap = ks / (CIN / 32); ci0 = 32 * (ks % (CIN / 32)) + 8 * g; }
}

// PROJ: instead of storing the activation, project it onto one output channel in fp32 (the 1 x 1 convolution that ends
// `pos_layers`, layers.py:439-472): proj_out[b][y][x] = sum_c pw[c] relu(conv[c] + bias[c]) + pb for y < Hv, x < Wv.
// MT_W: output-channel tiles (of 16) one wave computes.  The four waves split the 16 x 16 pixel tile as
// (COUT / 16 / MT_W) channel blocks x the remaining factor of row blocks: MT_W = COUT / 16 gives 4 rows x all channels
// per wave (each B fragment feeds MT_W MFMAs); MT_W = 1 with COUT = 32 gives 8 rows x 16 channels per wave, which
// keeps the register-resident weights of a 64-channel input at 72 VGPRs.
template <int CIN, int COUT, int MT_W, bool PROJ>
__global__ void __launch_bounds__(256, 2)
k_conv3x3(const uint16_t* __restrict__ in, const uint16_t* __restrict__ wfrag, const float* __restrict__ bias,
          uint16_t* __restrict__ out, uint16_t* __restrict__ pooled, int H, int W, int ostride, int ooff, int nchw,
          const float* __restrict__ pw, float pb, float* __restrict__ proj_out, int Hv, int Wv) {
  typedef ConvCfg<CIN> G;
  constexpr int MT = COUT / 16;
  constexpr int WM = MT / MT_W;        // waves along the output channels
  constexpr int RW = 16 / (4 / WM);    // rows of the tile per wave
  extern __shared__ uint16_t tile[];   // [18][18][PS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = W / 16;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int x0 = 16 * tx, y0 = 16 * ty;
  // weights -> registers (A fragments: lane l holds row l & 15 = output channel, k = 8 (l >> 4) .. + 7)
  const int mt0 = (wave % WM) * MT_W, row0 = (wave / WM) * RW;
  bf16x8 wf[G::KS][MT_W];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
    for (int mt = 0; mt < MT_W; ++mt) wf[ks][mt] = ((const bf16x8*)wfrag)[(ks * MT + mt0 + mt) * 64 + lane];
  // input tile + halo -> LDS, 16-byte chunks (8 channels); outside the image: zeros (SAME padding)
  {
    constexpr int CPP = CIN / 8;   // chunks per pixel
    const uint16_t* src = in + (size_t)b * H * W * CIN;
    for (int k = tid; k < G::TW * G::TW * CPP; k += 256) {
      const int p = k / CPP, ch = k - p * CPP;
      const int py = p / G::TW, px = p - py * G::TW;
      const int y = y0 + py - 1, x = x0 + px - 1;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (y >= 0 && y < H && x >= 0 && x < W) v = *(const uint4*)(src + ((size_t)y * W + x) * CIN + ch * 8);
      *(uint4*)(tile + p * G::PS + ch * 8) = v;
    }
  }
  __syncthreads();
  f32x4 acc[RW][MT_W];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int mt = 0; mt < MT_W; ++mt) acc[r][mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const int n = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) {
    int tap, ci0;
    k_of<CIN>(ks, g, tap, ci0);
    if (tap > 8) tap = 8;   // the padded tenth tap: its weights are zero, any valid address will do
    const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const bf16x8 xf = *(const bf16x8*)(tile + ((row0 + r + dy) * G::TW + n + dx) * G::PS + ci0);
#pragma unroll
      for (int mt = 0; mt < MT_W; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][mt], xf, acc[r][mt], 0, 0, 0);
    }
  }
  // epilogue.  D: lane holds column lane & 15 = pixel n, rows 4 g .. 4 g + 3 = output channels of tile mt
  if (PROJ) {
    static_assert(!PROJ || MT_W == MT, "the projection epilogue needs all channels in one wave");
    float part[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) part[r] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT_W; ++mt) {
      const int co = 16 * mt + 4 * g;
      const float4 bz = *(const float4*)(bias + co), wz = *(const float4*)(pw + co);
#pragma unroll
      for (int r = 0; r < RW; ++r)
        part[r] += (wz.x * fmaxf(acc[r][mt][0] + bz.x, 0.0f) + wz.y * fmaxf(acc[r][mt][1] + bz.y, 0.0f)) +
                   (wz.z * fmaxf(acc[r][mt][2] + bz.z, 0.0f) + wz.w * fmaxf(acc[r][mt][3] + bz.w, 0.0f));
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {   // the four lane groups hold the four channel quarters of one pixel
      part[r] = quarter_sum(part[r]);
      const int y = y0 + row0 + r, x = x0 + n;
      if (g == 0 && y < Hv && x < Wv) proj_out[((size_t)b * Hv + y) * Wv + x] = part[r] + pb;
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT_W; ++mt) {
    const int co = 16 * (mt0 + mt) + 4 * g;
    const float4 bz = *(const float4*)(bias + co);
    uint32_t lo[RW], hi[RW];   // packed bf16 pairs per row
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const float v0 = fmaxf(acc[r][mt][0] + bz.x, 0.0f), v1 = fmaxf(acc[r][mt][1] + bz.y, 0.0f);
      const float v2 = fmaxf(acc[r][mt][2] + bz.z, 0.0f), v3 = fmaxf(acc[r][mt][3] + bz.w, 0.0f);
      lo[r] = srl_pk_bf16(v0, v1); hi[r] = srl_pk_bf16(v2, v3);
      const int y = y0 + row0 + r, x = x0 + n;
      if (nchw) {
        uint16_t* o = out + ((size_t)b * COUT + co) * H * W + (size_t)y * W + x;
        o[0] = (uint16_t)(lo[r] & 0xffffu); o[(size_t)H * W] = (uint16_t)(lo[r] >> 16);
        o[(size_t)2 * H * W] = (uint16_t)(hi[r] & 0xffffu); o[(size_t)3 * H * W] = (uint16_t)(hi[r] >> 16);
      } else {
        *(uint2*)(out + (((size_t)b * H + y) * W + x) * ostride + ooff + co) = make_uint2(lo[r], hi[r]);
      }
    }
    if (pooled) {   // 2 x 2 max of the rounded values (bf16 bit patterns of non-negative numbers order like integers)
#pragma unroll
      for (int rp = 0; rp < RW / 2; ++rp) {
        uint32_t m[2] = {lo[2 * rp], hi[2 * rp]};
        const uint32_t o[2] = {lo[2 * rp + 1], hi[2 * rp + 1]};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          uint32_t a0 = m[q] & 0xffffu, a1 = m[q] >> 16, b0 = o[q] & 0xffffu, b1 = o[q] >> 16;
          a0 = a0 > b0 ? a0 : b0; a1 = a1 > b1 ? a1 : b1;
          // the neighbouring column: lane n ^ 1 (quad_perm [1, 0, 3, 2])
          const uint32_t mine = a0 | (a1 << 16);
          const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp((int)mine, (int)mine, 0xb1, 0xf, 0xf, false);
          uint32_t c0 = other & 0xffffu, c1 = other >> 16;
          a0 = a0 > c0 ? a0 : c0; a1 = a1 > c1 ? a1 : c1;
          m[q] = a0 | (a1 << 16);
        }
        if (!(n & 1)) {
          const int y2 = (y0 + row0) / 2 + rp, x2 = (x0 + n) / 2;
          *(uint2*)(pooled + (((size_t)b * (H / 2) + y2) * (W / 2) + x2) * COUT + co) = make_uint2(m[0], m[1]);
        }
      }
    }
  }
}

// 16-byte store of a finished float32 activation, non-temporal: at rollout batch sizes these tensors (0.25 - 2 GB per layer at
// 2,048 samples) are larger than the last-level cache, the lines only pass through on their way to HBM, and streaming them
// leaves the L2 to the loads (the fused first level at 2,048 x 128^2: 1,345 -> 1,102 us per launch, tools/ab_fused.py).
__device__ __forceinline__ void store_f4(float* p, float a, float b, float c, float d) {
  const f32x4 v = {a, b, c, d};
  __builtin_nontemporal_store(v, (f32x4*)p);
}

// Epilogue of the fp32-class kernels (k_conv3x3_x3, k_thin_conv3x3_x3).  D: lane holds column lane & 15 = pixel n, rows
// 4 g .. 4 g + 3 = output channels of tile mt.  Bias + ReLU, then 16-byte stores into a channel slice of a channels-last
// buffer, or channel-major output, optionally the 2 x 2 max-pooled tensor too; PROJ: the 1 x 1 projection instead.
template <int COUT, int MT_W, int RW, bool PROJ>
__device__ __forceinline__ void x3_epilogue(const f32x4 (&acc)[RW][MT_W], const float* __restrict__ bias, float* __restrict__ out,
                                            float* __restrict__ pooled, int H, int W, int ostride, int ooff, int nchw,
                                            const float* __restrict__ pw, float pb, float* __restrict__ proj_out, int Hv, int Wv,
                                            int b, int x0, int y0, int mt0, int row0, int n, int g) {
  constexpr int MT = COUT / 16;
  if (PROJ) {
    static_assert(!PROJ || MT_W == MT, "the projection epilogue needs all channels in one wave");
    float part[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) part[r] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT_W; ++mt) {
      const int co = 16 * mt + 4 * g;
      const float4 bz = *(const float4*)(bias + co), wz = *(const float4*)(pw + co);
#pragma unroll
      for (int r = 0; r < RW; ++r)
        part[r] += (wz.x * fmaxf(acc[r][mt][0] + bz.x, 0.0f) + wz.y * fmaxf(acc[r][mt][1] + bz.y, 0.0f)) +
                   (wz.z * fmaxf(acc[r][mt][2] + bz.z, 0.0f) + wz.w * fmaxf(acc[r][mt][3] + bz.w, 0.0f));
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      part[r] = quarter_sum(part[r]);
      const int y = y0 + row0 + r, x = x0 + n;
      if (g == 0 && y < Hv && x < Wv) proj_out[((size_t)b * Hv + y) * Wv + x] = part[r] + pb;
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT_W; ++mt) {
    const int co = 16 * (mt0 + mt) + 4 * g;
    const float4 bz = *(const float4*)(bias + co);
    float4 val[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      val[r] = make_float4(fmaxf(acc[r][mt][0] + bz.x, 0.0f), fmaxf(acc[r][mt][1] + bz.y, 0.0f),
                           fmaxf(acc[r][mt][2] + bz.z, 0.0f), fmaxf(acc[r][mt][3] + bz.w, 0.0f));
      const int y = y0 + row0 + r, x = x0 + n;
      if (nchw) {
        float* o = out + ((size_t)b * COUT + co) * H * W + (size_t)y * W + x;
        o[0] = val[r].x; o[(size_t)H * W] = val[r].y; o[(size_t)2 * H * W] = val[r].z; o[(size_t)3 * H * W] = val[r].w;
      } else {
        store_f4(out + (((size_t)b * H + y) * W + x) * ostride + ooff + co, val[r].x, val[r].y, val[r].z, val[r].w);
      }
    }
    if (pooled) {   // 2 x 2 max: rows in registers, the neighbouring column (lane n ^ 1) by a DPP quad permute
#pragma unroll
      for (int rp = 0; rp < RW / 2; ++rp) {
        float m[4] = {fmaxf(val[2 * rp].x, val[2 * rp + 1].x), fmaxf(val[2 * rp].y, val[2 * rp + 1].y),
                      fmaxf(val[2 * rp].z, val[2 * rp + 1].z), fmaxf(val[2 * rp].w, val[2 * rp + 1].w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float other = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m[q]), __float_as_int(m[q]), 0xb1, 0xf, 0xf, false));
          m[q] = fmaxf(m[q], other);
        }
        if (!(n & 1)) {
          const int y2 = (y0 + row0) / 2 + rp, x2 = (x0 + n) / 2;
          store_f4(pooled + (((size_t)b * (H / 2) + y2) * (W / 2) + x2) * COUT + co, m[0], m[1], m[2], m[3]);
        }
      }
    }
  }
}

// Stage one pass of an fp32-class kernel: the 18 x 18 pixel window (halo included, zero outside the H x W image) of CB float32
// channels at `src` (pixel stride `cstride` floats) into the hi / lo bf16 planes of the LDS tile.  The loads of BATCH
// thread-items are issued together — unconditionally, from clamped addresses, selected to zero afterwards — before the first
// is converted: written as a plain loop the compiler waited for each item's two loads before issuing the next's (six HBM
// round trips in a row per pass at CB = 32).
template <int CB, int BATCH>
__device__ __forceinline__ void stage_tile_x3(const float* __restrict__ src, int cstride, int H, int W, int x0, int y0,
                                              uint16_t* tile, int plane, int tid) {
  typedef ConvCfg<CB> G;
  constexpr int CPP = CB / 8;                       // 8-channel chunks per pixel
  constexpr int N = G::TW * G::TW * CPP, NIT = (N + 255) / 256;
#pragma unroll
  for (int i0 = 0; i0 < NIT; i0 += BATCH) {
    float4 a[BATCH], c[BATCH];
    bool ok[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      if (i0 + i >= NIT) continue;
      const int k = tid + 256 * (i0 + i), kk = k < N ? k : N - 1;
      const int p = kk / CPP, ch = kk - p * CPP;
      const int py = p / G::TW, px = p - py * G::TW;
      const int y = y0 + py - 1, x = x0 + px - 1;
      ok[i] = k < N && y >= 0 && y < H && x >= 0 && x < W;
      const float* q = src + ((size_t)(ok[i] ? y : 0) * W + (ok[i] ? x : 0)) * cstride + ch * 8;
      a[i] = *(const float4*)q; c[i] = *(const float4*)(q + 4);
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      if (i0 + i >= NIT) continue;
      const int k = tid + 256 * (i0 + i);
      if (k < N) {
        const int p = k / CPP, ch = k - p * CPP;
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float4 va = ok[i] ? a[i] : z, vc = ok[i] ? c[i] : z;
        uint32_t hi[4], lo[4];
        srl_split_bf16(va.x, va.y, hi[0], lo[0]); srl_split_bf16(va.z, va.w, hi[1], lo[1]);
        srl_split_bf16(vc.x, vc.y, hi[2], lo[2]); srl_split_bf16(vc.z, vc.w, hi[3], lo[3]);
        *(uint4*)(tile + p * G::PS + ch * 8) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *(uint4*)(tile + plane + p * G::PS + ch * 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
  }
}

// The MFMA loop of the fp32-class kernels over one staged pass (CB channels: hi plane at `tile`, lo plane PLANE elements on):
// acc[r][mt] += W x X over the KS K-steps, three products per fragment pair (small terms first: lo hi + hi lo, then the leading
// one).  The B fragments (16-byte LDS reads) are read ONE step (ks, r) ahead of their MFMAs and pinned there with
// sched_barrier: left alone the scheduler sinks every read to its first use and a step pays one LDS latency for its 3 MT_W
// MFMAs (the read-ahead form of k_conv3x3_gemm, csrc/conv_gemm.hip, where it was measured first).
template <int CB, int RW, int MT_W>
__device__ __forceinline__ void x3_mfma_pass(const uint16_t* tile, int plane, int row0, int n, int g,
                                             const bf16x8 (&wh)[ConvCfg<CB>::KS][MT_W], const bf16x8 (&wl)[ConvCfg<CB>::KS][MT_W],
                                             f32x4 (&acc)[RW][MT_W]) {
  typedef ConvCfg<CB> G;
  // element offset of the lane's fragment for K-step ks in tile row row0: the tap may depend on the lane (CB = 16: two taps
  // per step), the row r then adds a constant
  auto koff = [&](int ks) {
    int tap, ci0;
    k_of<CB>(ks, g, tap, ci0);
    if (tap > 8) tap = 8;   // the padded tenth tap: its weights are zero, any valid address will do
    const int dy = tap / 3, dx = tap - 3 * dy;
    return ((row0 + dy) * G::TW + n + dx) * G::PS + ci0;
  };
  int off = koff(0);
  bf16x8 xh = *(const bf16x8*)(tile + off), xl = *(const bf16x8*)(tile + plane + off);
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) {
    const int offn = ks + 1 < G::KS ? koff(ks + 1) : off;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      bf16x8 yh = xh, yl = xl;
      if (r + 1 < RW || ks + 1 < G::KS) {
        const int o = r + 1 < RW ? off + (r + 1) * G::TW * G::PS : offn;
        yh = *(const bf16x8*)(tile + o); yl = *(const bf16x8*)(tile + plane + o);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT_W; ++mt) {
        acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks][mt], xh, acc[r][mt], 0, 0, 0);
        acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks][mt], xl, acc[r][mt], 0, 0, 0);
        acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks][mt], xh, acc[r][mt], 0, 0, 0);
      }
      xh = yh; xl = yl;
    }
    off = offn;
  }
}

// The same convolution in fp32-class precision for the fp32 rollout (the reference's dtype): float32 channels-last in and
// out, every operand split into a high and a low bfloat16 part (x = hi + lo up to 2^-17 |x|) and the product taken as
// hi hi + hi lo + lo hi on the matrix cores with fp32 accumulation ("bf16x3": relative error ~2^-16 per product, the
// split the cross-correlation kernel uses) — a third of the bf16 MFMA rate, which these HBM-bound layers do not miss.
// The input tile is split once while it is staged (two bf16 planes in LDS), the weights arrive pre-split (hi fragments,
// then lo fragments) and stay in registers.  CIN in {16, 32, 64}; 64 input channels (the decoder's 32 + 32 concat
// buffer) go through in two passes of 32 — stage, load that half's weights, accumulate — so the tile stays at 52 KB
// of LDS and the weights at 144 VGPRs.
// PROJ: the fused 1 x 1 projection of `pos_layers`, as in k_conv3x3.
template <int CIN, int COUT, int MT_W, bool PROJ>
__global__ void __launch_bounds__(256, 2)
k_conv3x3_x3(const float* __restrict__ in, const uint16_t* __restrict__ wfrag, const float* __restrict__ bias,
             float* __restrict__ out, float* __restrict__ pooled, int H, int W, int ostride, int ooff, int nchw,
             const float* __restrict__ pw, float pb, float* __restrict__ proj_out, int Hv, int Wv) {
  constexpr int CB = CIN > 32 ? 32 : CIN;   // channels per pass
  constexpr int NP = CIN / CB;              // passes
  typedef ConvCfg<CB> G;
  constexpr int KST = G::KS * NP;           // K steps of the whole layer (fragment order: tap-major, 32-channel block minor)
  constexpr int MT = COUT / 16;
  constexpr int WM = MT / MT_W;
  constexpr int RW = 16 / (4 / WM);
  constexpr int PLANE = G::TW * G::TW * G::PS;
  extern __shared__ uint16_t tile[];   // [2][18][18][PS]: hi plane, lo plane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = W / 16;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int x0 = 16 * tx, y0 = 16 * ty;
  const int mt0 = (wave % WM) * MT_W, row0 = (wave / WM) * RW;
  const int n = lane & 15, g = lane >> 4;
  f32x4 acc[RW][MT_W];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int mt = 0; mt < MT_W; ++mt) acc[r][mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    bf16x8 wh[G::KS][MT_W], wl[G::KS][MT_W];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT_W; ++mt) {
        const int kg = ks * NP + ps;
        wh[ks][mt] = ((const bf16x8*)wfrag)[(kg * MT + mt0 + mt) * 64 + lane];
        wl[ks][mt] = ((const bf16x8*)wfrag)[((KST + kg) * MT + mt0 + mt) * 64 + lane];
      }
    if (ps) __syncthreads();   // the previous pass has read its tile
    {
      // (the weights above hold 8 KS MT_W registers: with 144 of them the window's loads go in two batches)
      constexpr int NIT = (G::TW * G::TW * (CB / 8) + 255) / 256;
      // (16-channel passes, three items per thread: batching measured 8 % slower at 16 -> 16, 128^2 — left item by item)
      stage_tile_x3<CB, (CB == 16 ? 1 : G::KS * MT_W >= 18 ? (NIT + 1) / 2 : NIT)>(in + (size_t)b * H * W * CIN + ps * CB, CIN, H, W, x0, y0, tile, PLANE, tid);
    }
    __syncthreads();
    x3_mfma_pass<CB, RW, MT_W>(tile, PLANE, row0, n, g, wh, wl, acc);
  }
  x3_epilogue<COUT, MT_W, RW, PROJ>(acc, bias, out, pooled, H, W, ostride, ooff, nchw, pw, pb, proj_out, Hv, Wv, b, x0, y0, mt0, row0, n, g);
}

// what one tap of the thin convolution loads: the pixel's CIN values in one register (pair)
template <int CIN, typename TIN> struct ThinRaw;
template <> struct ThinRaw<1, uint8_t> {
  typedef uint32_t type;
  static __device__ __forceinline__ type load(const uint8_t* p) { return *p; }
  static __device__ __forceinline__ float get(type r, int) { return (float)(r & 0xffu); }
};
template <> struct ThinRaw<2, uint8_t> {
  typedef uint32_t type;
  static __device__ __forceinline__ type load(const uint8_t* p) { return *(const uint16_t*)p; }
  static __device__ __forceinline__ float get(type r, int c) { return (float)((r >> (8 * c)) & 0xffu); }
};
template <> struct ThinRaw<1, float> {
  typedef float type;
  static __device__ __forceinline__ type load(const float* p) { return *p; }
  static __device__ __forceinline__ float get(type r, int) { return r; }
};
template <> struct ThinRaw<2, float> {
  typedef uint64_t type;
  static __device__ __forceinline__ type load(const float* p) { return *(const uint64_t*)p; }
  static __device__ __forceinline__ float get(type r, int c) { return __uint_as_float((uint32_t)(r >> (32 * c))); }
};

// NCO outputs of the thin layer for one pixel from its nine taps: relu(bias + sum_k w[k] v[k]), k = CIN tap + channel (the
// weights arrive packed [co][tap][channel]: a two-channel pixel is one 8-byte operand pair as it lies in memory).
// Uniform addresses: the weights and biases arrive by scalar loads and are SGPR operands of the FMAs; two partial sums over
// alternate k so that the FMAs pair up as v_pk_fma_f32 (weights w[k], w[k + 1] are adjacent SGPRs).  One statement of
// the arithmetic for k_conv3x3_thin and the fused k_thin_conv3x3_x3: their values are equal bit for bit.
template <int CIN, int NCO>
__device__ __forceinline__ void thin_outputs(const float (&v)[9 * CIN], const float* __restrict__ w, const float* __restrict__ bias,
                                             float (&r)[NCO]) {
#pragma unroll
  for (int co = 0; co < NCO; ++co) {
    const float* wk = w + co * CIN * 9;
    f32x2 a = {bias[co], 0.0f};
#pragma unroll
    for (int k = 0; k + 1 < 9 * CIN; k += 2) {
      const f32x2 wv = {wk[k], wk[k + 1]};
      const f32x2 xv = {v[k], v[k + 1]};
      a = __builtin_elementwise_fma(wv, xv, a);
    }
    float sum = a[0] + a[1];
    if ((9 * CIN) & 1) sum = fmaf(wk[9 * CIN - 1], v[9 * CIN - 1], sum);
    r[co] = fmaxf(sum, 0.0f);
  }
}

// 3 x 3 convolution + bias + ReLU from 1 or 2 input channels to 16 (the first layer of each U-Net and of
// `pos_layers`): K = 9 or 18 is too thin for the matrix cores, the layer is bound by its 32-byte-per-pixel output.
// One thread per output pixel, fp32 math, weights in SGPRs (scalar loads), bf16 channels-last output into a buffer
// of Hp x Wp pixels (>= H x W; the margin is left untouched: zero for the padded maps the MFMA kernel reads).
// TIN = uint8_t: the env's observation bytes, scaled by 1/255 here (models.py:144-147); float: as is.
// TOUT = uint16_t: bf16 output; float: the fp32 rollout (the arithmetic is fp32 either way).
template <int CIN, typename TIN, typename TOUT>
__global__ void __launch_bounds__(256)
k_conv3x3_thin(const TIN* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
               TOUT* __restrict__ out, int H, int W, int Hp, int Wp) {
  constexpr int WPP = 16 * sizeof(TOUT) / 4;   // 32-bit words per output pixel
  constexpr int NQ = WPP / 4;                  // 16-byte quads per pixel = lanes that share a pixel when storing
  constexpr int LS = WPP + 4;                  // LDS pixel stride in words (16-byte aligned, de-phased banks)
  __shared__ uint32_t xch[256 * LS];
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int y = p / W, x = p - y * W;          // p >= H W: every tap is out of range, the result is not stored
  const float scale = sizeof(TIN) == 1 ? 1.0f / 255.0f : 1.0f;
  // unconditional loads from clamped coordinates, all in flight together, zeroed afterwards (SAME padding); the two
  // bytes of a 2-channel uint8 pixel come as one 16-bit load
  typedef typename ThinRaw<CIN, TIN>::type RAW;
  RAW raw[9];
  bool ok[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    ok[t] = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
    raw[t] = ThinRaw<CIN, TIN>::load(in + ((size_t)b * H * W + (unsigned)(yc * W + xc)) * CIN);
  }
  // one fence for the nine values: keeps the loads out of the selects' branches without serialising them
  asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]),
               "+v"(raw[7]), "+v"(raw[8]));
  float v[9 * CIN];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CIN; ++c) v[CIN * t + c] = ok[t] ? ThinRaw<CIN, TIN>::get(raw[t], c) * scale : 0.0f;
  float r[16];
  thin_outputs<CIN, 16>(v, w, bias, r);
  // a thread's pixel is 32 or 64 contiguous bytes: exchanged through LDS so that every store instruction of a wave
  // writes one contiguous kilobyte (NQ adjacent lanes = the quads of one pixel)
  uint32_t* mine = xch + threadIdx.x * LS;
  if (sizeof(TOUT) == 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *(float4*)(mine + 4 * q) = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
  } else {
    uint32_t pk[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) pk[q] = srl_pk_bf16(r[2 * q], r[2 * q + 1]);
    *(uint4*)mine = make_uint4(pk[0], pk[1], pk[2], pk[3]); *(uint4*)(mine + 4) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
  }
  __syncthreads();
  const int wave0 = threadIdx.x & ~63, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int t = wave0 + i * (64 / NQ) + lane / NQ, q = lane % NQ;   // source thread (pixel) and quad
    const int pp = blockIdx.x * 256 + t;
    if (pp < H * W) {
      const int py = pp / W, px = pp - py * W;
      *(uint4*)((uint32_t*)(out + (((size_t)b * Hp + py) * Wp + px) * 16) + 4 * q) = *(const uint4*)(xch + t * LS + 4 * q);
    }
  }
}

// The thin layer and the 16 -> 16 layer behind it as ONE kernel, fp32-class (float32 rollout): the first two layers of each
// U-Net's encoder (uint8 observation -> 16 -> 16 [+ pooled]) and the whole of `pos_layers` (correlation map -> 16 -> 16 -> 1,
// layers.py:439-472).  As two kernels the 16-channel intermediate went out to HBM and came back (64 bytes per pixel each
// way against 1 - 8 bytes of input); here a workgroup keeps it in LDS: it stages the raw 20 x 20 input tile, evaluates the
// thin layer on the vector ALU for the 18 x 18 pixels the 3 x 3 taps of its 16 x 16 outputs reach (zero outside the
// H x W image: the second layer's SAME padding, and the zero margin of the padded map the two-kernel path used), splits
// the values into the hi / lo bf16 planes and runs k_conv3x3_x3<16, 16>'s MFMA loop and epilogue on them.  Same
// arithmetic per value as k_conv3x3_thin followed by k_conv3x3_x3: the outputs are equal bit for bit (GPU test).
// Thin-layer work items are (pixel, half of the 16 channels), dealt out wave by wave so that a wave's weights stay uniform
// (SGPR operands): five full waves of pixels per half, 8 outputs per thread and round; the four pixels left over (324 =
// 5 x 64 + 4) go to one more wave as (pixel, channel) per lane.
template <int CT, typename TIN, bool PROJ>
__global__ void __launch_bounds__(256, 4)
k_thin_conv3x3_x3(const TIN* __restrict__ in, const float* __restrict__ w1, const float* __restrict__ b1,
                  const uint16_t* __restrict__ wfrag, const float* __restrict__ bias, float* __restrict__ out,
                  float* __restrict__ pooled, int H, int W, int ostride, int ooff, int nchw, const float* __restrict__ pw, float pb,
                  float* __restrict__ proj_out) {
  typedef ConvCfg<16> G;
  constexpr int RT = G::TW + 2;                     // raw tile width: the halo of the halo
  constexpr int PLANE = G::TW * G::TW * G::PS;
  constexpr int RW = 4;
  extern __shared__ uint16_t tile[];                // [2][18][18][PS]: hi plane, lo plane
  __shared__ float rawt[RT * RT * CT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + 15) / 16;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int x0 = 16 * tx, y0 = 16 * ty;
  const int row0 = wave * RW;
  const int n = lane & 15, g = lane >> 4;
  bf16x8 wh[G::KS][1], wl[G::KS][1];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) {
    wh[ks][0] = ((const bf16x8*)wfrag)[ks * 64 + lane];
    wl[ks][0] = ((const bf16x8*)wfrag)[(G::KS + ks) * 64 + lane];
  }
  {
    const float scale = sizeof(TIN) == 1 ? 1.0f / 255.0f : 1.0f;
    const TIN* src = in + (size_t)b * H * W * CT;
    for (int k = tid; k < RT * RT; k += 256) {
      const int ry = k / RT, rx = k - ry * RT;
      const int y = y0 + ry - 2, x = x0 + rx - 2;
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      const typename ThinRaw<CT, TIN>::type raw = ThinRaw<CT, TIN>::load(src + (size_t)(ok ? y * W + x : 0) * CT);
#pragma unroll
      for (int c = 0; c < CT; ++c) rawt[k * CT + c] = ok ? ThinRaw<CT, TIN>::get(raw, c) * scale : 0.0f;
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int j = 0; j < 3; ++j) {
    const int wid = __builtin_amdgcn_readfirstlane(4 * j + wave);   // 0 .. 11
    if (wid < 10) {        // five full waves of pixels (0 .. 319) per channel half
      const int half = wid >= 5 ? 1 : 0;
      const int q = (wid - 5 * half) * 64 + lane;
      const int py = q / G::TW, px = q - py * G::TW;
      const int y = y0 + py - 1, x = x0 + px - 1;
      float v[9 * CT];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c) v[CT * t + c] = rawt[((py + t / 3) * RT + px + t % 3) * CT + c];
      float r[8];
      thin_outputs<CT, 8>(v, w1 + half * 8 * CT * 9, b1 + half * 8, r);
      const bool inside = y >= 0 && y < H && x >= 0 && x < W;
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) srl_split_bf16(inside ? r[2 * i] : 0.0f, inside ? r[2 * i + 1] : 0.0f, hi[i], lo[i]);
      *(uint4*)(tile + q * G::PS + 8 * half) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      *(uint4*)(tile + PLANE + q * G::PS + 8 * half) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    } else if (wid == 10) {   // the last four pixels (320 .. 323): one lane per (pixel, channel), weights per lane — the same
      const int q = 320 + (lane >> 4), co = lane & 15;   // two sums, term by term, as thin_outputs' packed FMAs
      const int py = q / G::TW, px = q - py * G::TW;
      const int y = y0 + py - 1, x = x0 + px - 1;
      const float* wk = w1 + co * CT * 9;
      float a0 = b1[co], a1 = 0.0f;
#pragma unroll
      for (int k = 0; k + 1 < 9 * CT; k += 2) {
        a0 = fmaf(wk[k], rawt[((py + (k / CT) / 3) * RT + px + (k / CT) % 3) * CT + k % CT], a0);
        a1 = fmaf(wk[k + 1], rawt[((py + ((k + 1) / CT) / 3) * RT + px + ((k + 1) / CT) % 3) * CT + (k + 1) % CT], a1);
      }
      float sum = a0 + a1;
      if ((9 * CT) & 1) sum = fmaf(wk[9 * CT - 1], rawt[((py + 2) * RT + px + 2) * CT + CT - 1], sum);
      const bool inside = y >= 0 && y < H && x >= 0 && x < W;
      uint32_t hi, lo;
      srl_split_bf16(inside ? fmaxf(sum, 0.0f) : 0.0f, 0.0f, hi, lo);
      tile[q * G::PS + co] = (uint16_t)hi;
      tile[PLANE + q * G::PS + co] = (uint16_t)lo;
    }
  }
  __syncthreads();
  f32x4 acc[RW][1];
#pragma unroll
  for (int r = 0; r < RW; ++r) acc[r][0] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  x3_mfma_pass<16, RW, 1>(tile, PLANE, row0, n, g, wh, wl, acc);
  x3_epilogue<16, 1, RW, PROJ>(acc, bias, out, pooled, H, W, ostride, ooff, nchw, pw, pb, proj_out, H, W, b, x0, y0, 0, row0, n, g);
}

// Transposed convolution 2 x 2, stride 2 (`up{i}` of layers.unet, layers.py:222-229) + bias + ReLU: every input pixel
// produces a 2 x 2 block of output pixels, out[2y+dy][2x+dx][co] = sum_ci in[y][x][ci] w[ci][co][dy][dx] — a plain
// GEMM per pixel with M = 4 COUT rows (dy, dx, co), no halo, so the B fragments (8 channels of one input pixel,
// 16 bytes) come straight from global memory.  Wave = 16 consecutive input pixels of a row x all M tiles; weights
// packed in A-fragment order and register-resident; output written into a channel slice of the channels-last
// concat buffer (8 bytes per lane: 4 consecutive channels of one output pixel).
template <int CIN, int COUT>
__global__ void __launch_bounds__(256, 2)
k_convt2x2(const uint16_t* __restrict__ in, const uint16_t* __restrict__ wfrag, const float* __restrict__ bias,
           uint16_t* __restrict__ out, int H, int W, int ostride, int ooff, long long ntiles) {
  constexpr int MT = 4 * COUT / 16, KS = CIN / 32;
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  bf16x8 wf[KS][MT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wf[ks][mt] = ((const bf16x8*)wfrag)[(ks * MT + mt) * 64 + lane];
  const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // 16 consecutive input pixels (W % 16 == 0)
  if (tile >= ntiles) return;
  const long long pix0 = tile * 16;
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bf16x8 xf = *(const bf16x8*)(in + (pix0 + n) * CIN + 32 * ks + 8 * g);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][mt], xf, acc[mt], 0, 0, 0);
  }
  const long long p = pix0 + n;
  const int x = (int)(p % W);
  const long long by = p / W;          // b * H + y
  const long long b = by / H; const int y = (int)(by - b * H);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = 16 * mt + 4 * g;     // row of the GEMM = (dy, dx, co)
    const int q = m / COUT, co = m - q * COUT, dy = q >> 1, dx = q & 1;
    const float4 bz = *(const float4*)(bias + co);
    const uint32_t lo = srl_pk_bf16(fmaxf(acc[mt][0] + bz.x, 0.0f), fmaxf(acc[mt][1] + bz.y, 0.0f));
    const uint32_t hi = srl_pk_bf16(fmaxf(acc[mt][2] + bz.z, 0.0f), fmaxf(acc[mt][3] + bz.w, 0.0f));
    *(uint2*)(out + (((b * 2 * H + 2 * y + dy) * 2 * W) + 2 * x + dx) * ostride + ooff + co) = make_uint2(lo, hi);
  }
}

// The transposed convolution in fp32-class precision (float32 in and out, bf16x3 products like k_conv3x3_x3): the lane's
// 8 input channels (32 bytes) are split into hi / lo parts in registers, the weights arrive pre-split (hi fragments, then
// lo fragments; 128 VGPRs for 64 -> 32); 16-byte stores of 4 consecutive output channels.
template <int CIN, int COUT>
__global__ void __launch_bounds__(256, 2)
k_convt2x2_x3(const float* __restrict__ in, const uint16_t* __restrict__ wfrag, const float* __restrict__ bias,
              float* __restrict__ out, int H, int W, int ostride, int ooff, long long ntiles) {
  constexpr int MT = 4 * COUT / 16, KS = CIN / 32;
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  bf16x8 wh[KS][MT], wl[KS][MT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      wh[ks][mt] = ((const bf16x8*)wfrag)[(ks * MT + mt) * 64 + lane];
      wl[ks][mt] = ((const bf16x8*)wfrag)[((KS + ks) * MT + mt) * 64 + lane];
    }
  const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // 16 consecutive input pixels (W % 16 == 0)
  if (tile >= ntiles) return;
  const long long pix0 = tile * 16;
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float* src = in + (pix0 + n) * CIN + 32 * ks + 8 * g;
    const float4 a = *(const float4*)src, c = *(const float4*)(src + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
    uint32_t ph[4], pl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) srl_split_bf16(v[2 * j], v[2 * j + 1], ph[j], pl[j]);
    const bf16x8 xh = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
    const bf16x8 xl = __builtin_bit_cast(bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks][mt], xh, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks][mt], xl, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks][mt], xh, acc[mt], 0, 0, 0);
    }
  }
  const long long p = pix0 + n;
  const int x = (int)(p % W);
  const long long by = p / W;          // b * H + y
  const long long b = by / H; const int y = (int)(by - b * H);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = 16 * mt + 4 * g;     // row of the GEMM = (dy, dx, co)
    const int q = m / COUT, co = m - q * COUT, dy = q >> 1, dx = q & 1;
    const float4 bz = *(const float4*)(bias + co);
    store_f4(out + (((b * 2 * H + 2 * y + dy) * 2 * W) + 2 * x + dx) * ostride + ooff + co, fmaxf(acc[mt][0] + bz.x, 0.0f),
             fmaxf(acc[mt][1] + bz.y, 0.0f), fmaxf(acc[mt][2] + bz.z, 0.0f), fmaxf(acc[mt][3] + bz.w, 0.0f));
  }
}

thread_local char c_err[256] = "";

template <int CIN, int COUT>
int launch(const void* in, const void* wfrag, const float* bias, void* out, void* pooled, int B, int H, int W, int ostride,
           int ooff, int nchw, hipStream_t st) {
  const size_t lds = sizeof(uint16_t) * ConvCfg<CIN>::TW * ConvCfg<CIN>::TW * ConvCfg<CIN>::PS;
  constexpr int MT_W = CIN >= 64 ? 1 : COUT / 16;
  hipLaunchKernelGGL((k_conv3x3<CIN, COUT, MT_W, false>), dim3((W / 16) * (H / 16), B), dim3(256), lds, st, (const uint16_t*)in,
                     (const uint16_t*)wfrag, bias, (uint16_t*)out, (uint16_t*)pooled, H, W, ostride, ooff, nchw,
                     (const float*)nullptr, 0.0f, (float*)nullptr, 0, 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(c_err, sizeof c_err, "srl_conv3x3_bias_relu: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

template <int CIN, int COUT>
int launch_x3(const float* in, const void* wfrag, const float* bias, float* out, float* pooled, int B, int H, int W, int ostride,
              int ooff, int nchw, hipStream_t st) {
  typedef ConvCfg<(CIN > 32 ? 32 : CIN)> G;   // 64 input channels: two passes of 32
  const size_t lds = 2 * sizeof(uint16_t) * G::TW * G::TW * G::PS;
  constexpr int MT_W = COUT / 16;
  hipLaunchKernelGGL((k_conv3x3_x3<CIN, COUT, MT_W, false>), dim3((W / 16) * (H / 16), B), dim3(256), lds, st, in, (const uint16_t*)wfrag,
                     bias, out, pooled, H, W, ostride, ooff, nchw, (const float*)nullptr, 0.0f, (float*)nullptr, 0, 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(c_err, sizeof c_err, "srl_conv3x3_bias_relu_f32: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

template <typename TOUT>
int launch_thin(const char* what, const void* in, int32_t in_dtype, const float* w, const float* bias, TOUT* o, int32_t B,
                int32_t H, int32_t W, int32_t cin, int32_t Hp, int32_t Wp, hipStream_t st) {
  if (!in || !w || !bias || !o || B < 1 || H < 1 || W < 1 || Hp < H || Wp < W || (cin != 1 && cin != 2) ||
      (in_dtype != 0 && in_dtype != 1)) {
    snprintf(c_err, sizeof c_err, "%s: bad arguments (cin in {1, 2}; in_dtype 0 = uint8 / 255, 1 = float32)", what);
    return 1;
  }
  const dim3 grid((H * W + 255) / 256, B), blk(256);
  if (cin == 1 && in_dtype == 0) hipLaunchKernelGGL((k_conv3x3_thin<1, uint8_t, TOUT>), grid, blk, 0, st, (const uint8_t*)in, w, bias, o, H, W, Hp, Wp);
  else if (cin == 1) hipLaunchKernelGGL((k_conv3x3_thin<1, float, TOUT>), grid, blk, 0, st, (const float*)in, w, bias, o, H, W, Hp, Wp);
  else if (in_dtype == 0) hipLaunchKernelGGL((k_conv3x3_thin<2, uint8_t, TOUT>), grid, blk, 0, st, (const uint8_t*)in, w, bias, o, H, W, Hp, Wp);
  else hipLaunchKernelGGL((k_conv3x3_thin<2, float, TOUT>), grid, blk, 0, st, (const float*)in, w, bias, o, H, W, Hp, Wp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(c_err, sizeof c_err, "%s: %s", what, hipGetErrorString(e)); return 2; }
  return 0;
}

template <bool PROJ>
int launch_thin_conv(const char* what, const void* in, int32_t in_dtype, int32_t cin, const float* w1, const float* b1, const void* wfrag,
                     const float* bias, float* out, float* pooled, int B, int H, int W, int ostride, int ooff, int nchw,
                     const float* pw, float pb, float* proj_out, hipStream_t st) {
  const size_t lds = 2 * sizeof(uint16_t) * ConvCfg<16>::TW * ConvCfg<16>::TW * ConvCfg<16>::PS;
  const dim3 grid(((W + 15) / 16) * ((H + 15) / 16), B), blk(256);
  const uint16_t* wf = (const uint16_t*)wfrag;
  if (cin == 1 && in_dtype == 0)
    hipLaunchKernelGGL((k_thin_conv3x3_x3<1, uint8_t, PROJ>), grid, blk, lds, st, (const uint8_t*)in, w1, b1, wf, bias, out, pooled, H, W, ostride, ooff, nchw, pw, pb, proj_out);
  else if (cin == 1)
    hipLaunchKernelGGL((k_thin_conv3x3_x3<1, float, PROJ>), grid, blk, lds, st, (const float*)in, w1, b1, wf, bias, out, pooled, H, W, ostride, ooff, nchw, pw, pb, proj_out);
  else if (in_dtype == 0)
    hipLaunchKernelGGL((k_thin_conv3x3_x3<2, uint8_t, PROJ>), grid, blk, lds, st, (const uint8_t*)in, w1, b1, wf, bias, out, pooled, H, W, ostride, ooff, nchw, pw, pb, proj_out);
  else
    hipLaunchKernelGGL((k_thin_conv3x3_x3<2, float, PROJ>), grid, blk, lds, st, (const float*)in, w1, b1, wf, bias, out, pooled, H, W, ostride, ooff, nchw, pw, pb, proj_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(c_err, sizeof c_err, "%s: %s", what, hipGetErrorString(e)); return 2; }
  return 0;
}

}  // namespace

extern "C" {

const char* srl_conv_last_error(void) { return c_err; }

int srl_conv3x3_bias_relu_f32(const float* in, const void* wfrag, const float* bias, float* out, float* pooled, int32_t B,
                              int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset,
                              int32_t nchw, void* stream) {
  if (!in || !wfrag || !bias || !out || B < 1 || H < 16 || W < 16 || H % 16 || W % 16 || out_stride % 4 || out_offset % 4 ||
      srl_conv3x3_wfrag_elems(cin, cout) < 0 || (pooled && nchw)) {
    snprintf(c_err, sizeof c_err, "srl_conv3x3_bias_relu_f32: bad arguments (H, W multiples of 16; cin in {16, 32, 64}, cout in {16, 32})");
    return 1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (cin == 16 && cout == 16) return launch_x3<16, 16>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  if (cin == 16 && cout == 32) return launch_x3<16, 32>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  if (cin == 32 && cout == 16) return launch_x3<32, 16>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  if (cin == 32 && cout == 32) return launch_x3<32, 32>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  if (cin == 64 && cout == 16) return launch_x3<64, 16>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  return launch_x3<64, 32>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
}

int32_t srl_conv3x3_wfrag_elems(int32_t cin, int32_t cout) {
  if ((cin != 16 && cin != 32 && cin != 64) || (cout != 16 && cout != 32)) return -1;
  return (cin == 16 ? 5 : 9 * (cin / 32)) * (cout / 16) * 64 * 8;
}

int srl_conv3x3_bias_relu(const void* in, const void* wfrag, const float* bias, void* out, void* pooled, int32_t B,
                          int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset,
                          int32_t nchw, void* stream) {
  if (!in || !wfrag || !bias || !out || B < 1 || H < 16 || W < 16 || H % 16 || W % 16 || out_stride % 4 || out_offset % 4 ||
      srl_conv3x3_wfrag_elems(cin, cout) < 0 || (pooled && nchw)) {
    snprintf(c_err, sizeof c_err, "srl_conv3x3_bias_relu: bad arguments (H, W multiples of 16; cin in {16, 32, 64}, cout in {16, 32})");
    return 1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (cin == 16 && cout == 16) return launch<16, 16>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  if (cin == 16 && cout == 32) return launch<16, 32>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  if (cin == 32 && cout == 16) return launch<32, 16>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  if (cin == 32 && cout == 32) return launch<32, 32>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  if (cin == 64 && cout == 16) return launch<64, 16>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
  return launch<64, 32>(in, wfrag, bias, out, pooled, B, H, W, out_stride, out_offset, nchw, st);
}

int srl_conv3x3_thin(const void* in, int32_t in_dtype, const float* w, const float* bias, void* out, int32_t B, int32_t H,
                     int32_t W, int32_t cin, int32_t Hp, int32_t Wp, void* stream) {
  return launch_thin("srl_conv3x3_thin", in, in_dtype, w, bias, (uint16_t*)out, B, H, W, cin, Hp, Wp, (hipStream_t)stream);
}

int srl_conv3x3_thin_f32(const void* in, int32_t in_dtype, const float* w, const float* bias, float* out, int32_t B, int32_t H,
                         int32_t W, int32_t cin, int32_t Hp, int32_t Wp, void* stream) {
  return launch_thin("srl_conv3x3_thin_f32", in, in_dtype, w, bias, out, B, H, W, cin, Hp, Wp, (hipStream_t)stream);
}

int32_t srl_convt2x2_wfrag_elems(int32_t cin, int32_t cout) {
  // (128 -> 64 and 256 -> 128: the same fragment order, consumed by srl_convt2x2_gemm_bias_relu, csrc/conv_gemm.hip)
  if (!((cin == 32 && cout == 16) || (cin == 64 && cout == 32) || (cin == 128 && cout == 64) || (cin == 256 && cout == 128))) return -1;
  return (cin / 32) * (4 * cout / 16) * 64 * 8;
}

int srl_convt2x2_bias_relu(const void* in, const void* wfrag, const float* bias, void* out, int32_t B, int32_t H, int32_t W,
                           int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset, void* stream) {
  if (!in || !wfrag || !bias || !out || B < 1 || H < 1 || W < 16 || W % 16 || out_stride % 4 || out_offset % 4 ||
      !((cin == 32 && cout == 16) || (cin == 64 && cout == 32))) {
    snprintf(c_err, sizeof c_err, "srl_convt2x2_bias_relu: bad arguments (W a multiple of 16; 32 -> 16 or 64 -> 32 channels)");
    return 1;
  }
  const long long ntiles = (long long)B * H * W / 16;
  const dim3 grid((unsigned)((ntiles + 3) / 4)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (cin == 32) hipLaunchKernelGGL((k_convt2x2<32, 16>), grid, blk, 0, st, (const uint16_t*)in, (const uint16_t*)wfrag, bias, (uint16_t*)out, H, W, out_stride, out_offset, ntiles);
  else hipLaunchKernelGGL((k_convt2x2<64, 32>), grid, blk, 0, st, (const uint16_t*)in, (const uint16_t*)wfrag, bias, (uint16_t*)out, H, W, out_stride, out_offset, ntiles);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(c_err, sizeof c_err, "srl_convt2x2_bias_relu: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

int srl_convt2x2_bias_relu_f32(const float* in, const void* wfrag, const float* bias, float* out, int32_t B, int32_t H, int32_t W,
                               int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset, void* stream) {
  if (!in || !wfrag || !bias || !out || B < 1 || H < 1 || W < 16 || W % 16 || out_stride % 4 || out_offset % 4 ||
      !((cin == 32 && cout == 16) || (cin == 64 && cout == 32))) {
    snprintf(c_err, sizeof c_err, "srl_convt2x2_bias_relu_f32: bad arguments (W a multiple of 16; 32 -> 16 or 64 -> 32 channels)");
    return 1;
  }
  const long long ntiles = (long long)B * H * W / 16;
  const dim3 grid((unsigned)((ntiles + 3) / 4)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (cin == 32) hipLaunchKernelGGL((k_convt2x2_x3<32, 16>), grid, blk, 0, st, in, (const uint16_t*)wfrag, bias, out, H, W, out_stride, out_offset, ntiles);
  else hipLaunchKernelGGL((k_convt2x2_x3<64, 32>), grid, blk, 0, st, in, (const uint16_t*)wfrag, bias, out, H, W, out_stride, out_offset, ntiles);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(c_err, sizeof c_err, "srl_convt2x2_bias_relu_f32: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

int srl_conv3x3_relu_project_f32(const float* in, const void* wfrag, const float* bias, const float* proj_w, float proj_b,
                                 float* out, int32_t B, int32_t H, int32_t W, int32_t Hv, int32_t Wv, void* stream) {
  if (!in || !wfrag || !bias || !proj_w || !out || B < 1 || H < 16 || W < 16 || H % 16 || W % 16 || Hv < 1 || Hv > H ||
      Wv < 1 || Wv > W) {
    snprintf(c_err, sizeof c_err, "srl_conv3x3_relu_project_f32: bad arguments");
    return 1;
  }
  const size_t lds = 2 * sizeof(uint16_t) * ConvCfg<16>::TW * ConvCfg<16>::TW * ConvCfg<16>::PS;
  hipLaunchKernelGGL((k_conv3x3_x3<16, 16, 1, true>), dim3((W / 16) * (H / 16), B), dim3(256), lds, (hipStream_t)stream, in,
                     (const uint16_t*)wfrag, bias, (float*)nullptr, (float*)nullptr, H, W, 16, 0, 0, proj_w, proj_b, out, Hv, Wv);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(c_err, sizeof c_err, "srl_conv3x3_relu_project_f32: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

int srl_thin_conv3x3_bias_relu_f32(const void* in, int32_t in_dtype, int32_t cin, const float* w1, const float* b1, const void* wfrag,
                                   const float* bias, float* out, float* pooled, int32_t B, int32_t H, int32_t W,
                                   int32_t out_stride, int32_t out_offset, int32_t nchw, void* stream) {
  if (!in || !w1 || !b1 || !wfrag || !bias || !out || B < 1 || H < 16 || W < 16 || H % 16 || W % 16 || out_stride % 4 ||
      out_offset % 4 || (cin != 1 && cin != 2) || (in_dtype != 0 && in_dtype != 1) || (pooled && nchw)) {
    snprintf(c_err, sizeof c_err, "srl_thin_conv3x3_bias_relu_f32: bad arguments (H, W multiples of 16; cin in {1, 2}; in_dtype 0 = uint8 / 255, 1 = float32)");
    return 1;
  }
  return launch_thin_conv<false>("srl_thin_conv3x3_bias_relu_f32", in, in_dtype, cin, w1, b1, wfrag, bias, out, pooled, B, H, W,
                                 out_stride, out_offset, nchw, nullptr, 0.0f, nullptr, (hipStream_t)stream);
}

int srl_thin_conv3x3_relu_project_f32(const float* in, const float* w1, const float* b1, const void* wfrag, const float* bias,
                                      const float* proj_w, float proj_b, float* out, int32_t B, int32_t H, int32_t W, void* stream) {
  if (!in || !w1 || !b1 || !wfrag || !bias || !proj_w || !out || B < 1 || H < 1 || W < 1) {
    snprintf(c_err, sizeof c_err, "srl_thin_conv3x3_relu_project_f32: bad arguments");
    return 1;
  }
  return launch_thin_conv<true>("srl_thin_conv3x3_relu_project_f32", in, 1, 1, w1, b1, wfrag, bias, nullptr, nullptr, B, H, W, 16, 0, 0,
                                proj_w, proj_b, out, (hipStream_t)stream);
}

int srl_conv3x3_relu_project(const void* in, const void* wfrag, const float* bias, const float* proj_w, float proj_b,
                             float* out, int32_t B, int32_t H, int32_t W, int32_t Hv, int32_t Wv, void* stream) {
  if (!in || !wfrag || !bias || !proj_w || !out || B < 1 || H < 16 || W < 16 || H % 16 || W % 16 || Hv < 1 || Hv > H ||
      Wv < 1 || Wv > W) {
    snprintf(c_err, sizeof c_err, "srl_conv3x3_relu_project: bad arguments");
    return 1;
  }
  const size_t lds = sizeof(uint16_t) * ConvCfg<16>::TW * ConvCfg<16>::TW * ConvCfg<16>::PS;
  hipLaunchKernelGGL((k_conv3x3<16, 16, 1, true>), dim3((W / 16) * (H / 16), B), dim3(256), lds, (hipStream_t)stream,
                     (const uint16_t*)in, (const uint16_t*)wfrag, bias, (uint16_t*)nullptr, (uint16_t*)nullptr, H, W, 16, 0, 0,
                     proj_w, proj_b, out, Hv, Wv);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(c_err, sizeof c_err, "srl_conv3x3_relu_project: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

}  // extern "C"
