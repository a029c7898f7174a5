// conv_gemm.hip — 3 x 3 convolution (stride 1, SAME) + bias + ReLU of the U-Nets' deep levels (64 / 128 / 256 output
// channels at 32^2 / 16^2 / 8^2) as an implicit GEMM on the matrix cores, inference only, gfx950.  Replaces
// Conv2D(f, 3, padding='same', activation='relu') of `layers.unet` (stackrl/nets/layers.py:135-259) where the weights
// no longer fit a wave's registers (csrc/conv_mfma.hip holds the 16- / 32-output-channel layers): these layers are
// compute-bound GEMMs, D[cout][pixel] += W[cout][k] X[k][pixel] with k = (32-channel block, tap), K = 9 CIN = 288 .. 2,304.
//
//   workgroup = 4 waves; a wave computes 64 output channels x 128 pixels (4 x 8 accumulator tiles of
//   v_mfma_f32_16x16x32_bf16, 128 VGPRs).  COUT / 64 waves lie along the channels, the others along the pixels, so the
//   workgroup's pixel tile is 512 / 256 / 128 pixels at COUT = 64 / 128 / 256: 16 rows of a 32^2 map, one 16^2 map, two
//   8^2 maps.
//   K loop: for every block of 32 input channels the input tile + halo (zero padded at the border) is staged in LDS once
//   (64 B per pixel, stride 80 B: conflict-free 16-byte reads) and serves the nine taps; a K step (one tap x 32 channels)
//   reads 8 B fragments from LDS and feeds 32 MFMAs.  The A fragments (weights, pre-packed in fragment order, L2-resident:
//   at most 1.2 MB per layer) come straight from global memory into registers, one K step ahead — no LDS traffic and no
//   barrier per tap.
//   epilogue: bias + ReLU on the 4 consecutive channels a lane holds, 8-byte stores into a channel slice of a
//   channels-last buffer.
// bf16 in / out with fp32 accumulation; the fp32-class variant (float32 in / out, bf16x3 products) splits the staged
// tile into hi / lo planes like k_conv3x3_x3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stackrl_qnet.h"
#include "srl_bf16.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


template <int COUT, int W, bool X3 = false>
struct GemmCfg {
  static constexpr int WM = COUT / 64;             // waves along the output channels
  static constexpr int WN = 4 / WM;                // waves along the pixels
  static constexpr int PXT = 128 * WN;             // pixels per workgroup
  static constexpr int NI = PXT >= W * W ? PXT / (W * W) : 1;   // whole maps per workgroup (8^2: two)
  static constexpr int RT = NI > 1 ? W : PXT / W;  // rows of a map per workgroup
  static constexpr int TH = RT + 2, TWD = W + 2;   // tile incl. halo, per map
  // LDS pixel stride in bf16 elements: 40 (80 B for the 64 B of a pixel's 32 channels: the sixteen lanes of a 16-byte read land
  // in sixteen different bank quads), or — SWZ: the fp32-class 64-channel layers at 32^2, whose two planes of 18 x 34 pixels
  // would be 98 KB and leave a CU to ONE workgroup, i.e. one wave per SIMD with nothing to run under its staging — 32 with the
  // pixel's four 16-byte chunks permuted by its column, slot = chunk ^ ((column >> 2) & 3): columns c, c + 4, c + 8, c + 12
  // share their sixteen banks and now use different quads of them.  78 KB, two workgroups per CU.
  static constexpr bool SWZ = X3 && COUT == 64 && W == 32;
  static constexpr int PS = SWZ ? 32 : 40;
  static constexpr int TILE = NI * TH * TWD * PS;  // elements per plane
  // element offset of N tile t's pixel relative to N tile 0's, the same for every lane
  static constexpr int tile_delta(int t) {
    return (((16 * t) / (RT * W) * TH + ((16 * t) / W) % RT) * TWD + (16 * t) % W) * PS;
  }
  static_assert(COUT == 64 || COUT == 128 || COUT == 256, "COUT");
  static_assert(PXT % W == 0 && (NI == 1 || PXT == NI * W * W), "pixel tile");
};

// X3: float32 tensors, bf16x3 products (hi hi + hi lo + lo hi); else bf16 tensors
template <int CIN, int COUT, int W, bool X3>
__global__ void __launch_bounds__(256, 2)
k_conv3x3_gemm(const void* __restrict__ in_, const uint16_t* __restrict__ wfrag, const float* __restrict__ bias,
               void* __restrict__ out_, int ostride, int ooff) {
  typedef GemmCfg<COUT, W, X3> G;
  constexpr int NCB = CIN / 32, MT = COUT / 16;
  extern __shared__ uint16_t tile[];               // [1 or (X3) 2 planes: hi, lo][NI][TH][TWD][PS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % G::WM, wn = wave / G::WM;
  const int n = lane & 15, g = lane >> 4;
  // the workgroup's pixels: maps [img0, img0 + NI), rows [row0, row0 + RT) of each
  constexpr int PARTS = (W * W + G::PXT - 1) / G::PXT;   // workgroups per map (when NI == 1)
  const int img0 = G::NI > 1 ? blockIdx.x * G::NI : blockIdx.x / PARTS;
  const int row0 = G::NI > 1 ? 0 : (blockIdx.x % PARTS) * G::RT;
  // LDS offset (elements) of the top-left tap of the lane's pixel in N tile 0 (N tile t = 16 consecutive pixels of the wave's
  // 128, row-major over the workgroup's pixel tile).  Tile t lies a whole number of images / rows / 16-column halves further
  // on for every lane (16 divides the row length or is a multiple of it), so its offset is this one plus a constant,
  // G::tile_delta(t): an immediate of the LDS read instead of a register per tile.
  int poff0;
  {
    const int p = wn * 128 + n;                    // pixel within the workgroup tile
    const int im = p / (G::RT * W), r = (p / W) % G::RT, c = p % W;
    poff0 = ((im * G::TH + r) * G::TWD + c) * G::PS + (G::SWZ ? 0 : 8 * g);
  }
  // SWZ: the lane's chunk slot for a tap in tile column c + dx; c = n + 16 (t & 1), so the key ((c + dx) >> 2) & 3 depends on
  // n and dx alone
  int sw[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) sw[dx] = G::SWZ ? 8 * (g ^ (((n + dx) >> 2) & 3)) : 0;
  f32x4 acc[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[mt][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  // A fragments of K step (cb, tap): wfrag[((cb * 9 + tap) * MT + mt) * 64 + lane]; X3: hi set, then lo set
  const bf16x8* wf = (const bf16x8*)wfrag + (4 * wm) * 64 + lane;
  constexpr size_t LO = (size_t)NCB * 9 * MT * 64;
  for (int cb = 0; cb < NCB; ++cb) {
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      ah[mt] = wf[((size_t)(cb * 9) * MT + mt) * 64];
      if (X3) al[mt] = wf[LO + ((size_t)(cb * 9) * MT + mt) * 64];
    }
    if (cb) __syncthreads();                       // the previous block's taps have read the tile
    {
      constexpr int NPX = G::NI * G::TH * G::TWD;  // pixels incl. halo; 4 chunks of 8 channels each
      for (int k = tid; k < NPX * 4; k += 256) {
        const int p = k >> 2, ch = k & 3;
        const int im = p / (G::TH * G::TWD), rr = (p / G::TWD) % G::TH, cc = p % G::TWD;
        const int y = row0 + rr - 1, x = cc - 1;
        const bool ok = y >= 0 && y < W && x >= 0 && x < W;
        const size_t src = (((size_t)(img0 + im) * W + (ok ? y : 0)) * W + (ok ? x : 0)) * CIN + cb * 32 + ch * 8;
        if (X3) {
          const float* s = (const float*)in_ + src;
          float4 a = *(const float4*)s, c = *(const float4*)(s + 4);
          if (!ok) { a = make_float4(0.0f, 0.0f, 0.0f, 0.0f); c = a; }
          const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
          uint32_t hi[4], lo[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) srl_split_bf16(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
          const int slot = G::SWZ ? (ch ^ ((cc >> 2) & 3)) : ch;
          *(uint4*)(tile + p * G::PS + slot * 8) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
          *(uint4*)(tile + G::TILE + p * G::PS + slot * 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        } else {
          uint4 v = *(const uint4*)((const uint16_t*)in_ + src);
          if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
          *(uint4*)(tile + p * G::PS + ch * 8) = v;
        }
      }
    }
    __syncthreads();
    bf16x8 pxh = {0, 0, 0, 0, 0, 0, 0, 0}, pxl = pxh;   // the fragments read ahead for the next tap's first tile
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      bf16x8 nh[4], nl[4];
      if (tap < 8) {                               // next tap's weights, in flight during this tap's MFMAs
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          nh[mt] = wf[((size_t)(cb * 9 + tap + 1) * MT + mt) * 64];
          if (X3) nl[mt] = wf[LO + ((size_t)(cb * 9 + tap + 1) * MT + mt) * 64];
        }
      }
      // B fragments one step ahead: tile t + 1's (or the next tap's first) LDS reads are in flight during tile t's MFMAs
      bf16x8 xh, xl;
      if (tap == 0) {
        const int xoff = poff0 + (G::SWZ ? sw[0] : 0);
        xh = *(const bf16x8*)(tile + xoff);
        if (X3) xl = *(const bf16x8*)(tile + G::TILE + xoff);
      } else {
        xh = pxh; xl = pxl;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        bf16x8 yh = xh, yl = xl;
        if (t < 7 || tap < 8) {
          const int tn = t < 7 ? t + 1 : 0, tp = t < 7 ? tap : tap + 1;
          const int xoff = poff0 + G::tile_delta(tn) + ((tp / 3) * G::TWD + tp % 3) * G::PS + (G::SWZ ? sw[tp % 3] : 0);
          yh = *(const bf16x8*)(tile + xoff);
          if (X3) yl = *(const bf16x8*)(tile + G::TILE + xoff);
        }
        // the reads stay here: issued before this tile's MFMAs (left alone the scheduler sinks them to their first use, one
        // LDS latency exposed per tile) and not earlier than the previous tile's (hoisted further they spill)
        __builtin_amdgcn_sched_barrier(0);
        if (X3) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], xh, acc[mt][t], 0, 0, 0);
            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], xl, acc[mt][t], 0, 0, 0);
            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], xh, acc[mt][t], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], xh, acc[mt][t], 0, 0, 0);
        }
        xh = yh; xl = yl;
      }
      pxh = xh; pxl = xl;
      if (tap < 8) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { ah[mt] = nh[mt]; if (X3) al[mt] = nl[mt]; }
      }
    }
  }
  // epilogue.  D: lane holds column n = pixel, rows 4 g .. 4 g + 3 = output channels of tile mt
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int co = 64 * wm + 16 * mt + 4 * g;
    const float4 bz = *(const float4*)(bias + co);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int p = wn * 128 + t * 16 + n;
      const int im = p / (G::RT * W), r = (p / W) % G::RT, c = p % W;
      const size_t pix = ((size_t)(img0 + im) * W + row0 + r) * W + c;
      const float v0 = fmaxf(acc[mt][t][0] + bz.x, 0.0f), v1 = fmaxf(acc[mt][t][1] + bz.y, 0.0f);
      const float v2 = fmaxf(acc[mt][t][2] + bz.z, 0.0f), v3 = fmaxf(acc[mt][t][3] + bz.w, 0.0f);
      if (X3) *(float4*)((float*)out_ + pix * ostride + ooff + co) = make_float4(v0, v1, v2, v3);
      else *(uint2*)((uint16_t*)out_ + pix * ostride + ooff + co) =
             make_uint2(srl_pk_bf16(v0, v1), srl_pk_bf16(v2, v3));
    }
  }
}

// Transposed convolution 2 x 2, stride 2 (`up{i}` of layers.unet, layers.py:222-229) + bias + ReLU of the deep levels
// (64 / 128 output channels) as a GEMM: every input pixel produces its 2 x 2 block of output pixels,
// D[(dy, dx, co)][pixel] = sum_ci W[ci][co][dy][dx] X[pixel][ci] — M = 4 COUT rows, K = CIN, no halo, any map size.
// A wave owns 64 rows x 128 consecutive input pixels (of the flattened [B][H][W] order); the four waves of a workgroup
// take four row blocks (COUT = 128: two workgroups per pixel tile).  K has only CIN / 32 = 2 .. 8 steps, so nothing is
// staged: the A fragments (pre-packed, L2-resident) and the B fragments (16 or, fp32-class, 32 bytes of one pixel's
// channels, shared by the workgroup's waves through L1) come straight from global memory.  Output: the channel slice of
// the concat buffer, 4 consecutive channels of one output pixel per lane and tile.
template <int CIN, int COUT, bool X3>
__global__ void __launch_bounds__(256, 2)
k_convt2x2_gemm(const void* __restrict__ in_, const uint16_t* __restrict__ wfrag, const float* __restrict__ bias,
                void* __restrict__ out_, int H, int W, int ostride, int ooff, long long npix) {
  constexpr int KS = CIN / 32, MT = 4 * COUT / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
  const int mb = blockIdx.y * 4 + wave;                 // 64-row block of the 4 COUT rows
  const long long p0 = (long long)blockIdx.x * 128;
  f32x4 acc[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[mt][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const bf16x8* wf = (const bf16x8*)wfrag + (4 * mb) * 64 + lane;
  constexpr size_t LO = (size_t)KS * MT * 64;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      ah[mt] = wf[((size_t)ks * MT + mt) * 64];
      if (X3) al[mt] = wf[LO + ((size_t)ks * MT + mt) * 64];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      long long p = p0 + 16 * t + n;
      if (p >= npix) p = npix - 1;                      // (a partial last tile: clamped loads, stores skipped)
      bf16x8 xh, xl;
      if (X3) {
        const float* s = (const float*)in_ + p * CIN + 32 * ks + 8 * g;
        const float4 a = *(const float4*)s, c = *(const float4*)(s + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        uint32_t ph[4], pl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) srl_split_bf16(v[2 * j], v[2 * j + 1], ph[j], pl[j]);
        xh = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
        xl = __builtin_bit_cast(bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
      } else {
        xh = *(const bf16x8*)((const uint16_t*)in_ + p * CIN + 32 * ks + 8 * g);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (X3) {
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], xh, acc[mt][t], 0, 0, 0);
          acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], xl, acc[mt][t], 0, 0, 0);
        }
        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], xh, acc[mt][t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const long long p = p0 + 16 * t + n;
    if (p >= npix) continue;
    const int x = (int)(p % W);
    const long long by = p / W;                         // b * H + y
    const long long b = by / H; const int y = (int)(by - b * H);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = 64 * mb + 16 * mt + 4 * g;          // row of the GEMM = (dy, dx, co)
      const int q = m / COUT, co = m - q * COUT, dy = q >> 1, dx = q & 1;
      const float4 bz = *(const float4*)(bias + co);
      const float v0 = fmaxf(acc[mt][t][0] + bz.x, 0.0f), v1 = fmaxf(acc[mt][t][1] + bz.y, 0.0f);
      const float v2 = fmaxf(acc[mt][t][2] + bz.z, 0.0f), v3 = fmaxf(acc[mt][t][3] + bz.w, 0.0f);
      const size_t o = (((size_t)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * ostride + ooff + co;
      if (X3) *(float4*)((float*)out_ + o) = make_float4(v0, v1, v2, v3);
      else *(uint2*)((uint16_t*)out_ + o) = make_uint2(srl_pk_bf16(v0, v1), srl_pk_bf16(v2, v3));
    }
  }
}

thread_local char gm_err[256] = "";

template <int CIN, int COUT, int W, bool X3>
int launch_gemm(const void* in, const void* wfrag, const float* bias, void* out, int B, int ostride, int ooff, hipStream_t st) {
  typedef GemmCfg<COUT, W, X3> G;
  const size_t lds = sizeof(uint16_t) * G::TILE * (X3 ? 2 : 1);
  const int nwg = G::NI > 1 || G::PXT == W * W ? B / G::NI : B * ((W * W) / G::PXT);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)k_conv3x3_gemm<CIN, COUT, W, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((k_conv3x3_gemm<CIN, COUT, W, X3>), dim3(nwg), dim3(256), lds, st, in, (const uint16_t*)wfrag, bias, out,
                     ostride, ooff);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(gm_err, sizeof gm_err, "srl_conv3x3_gemm: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

template <bool X3>
int dispatch(const void* in, const void* wfrag, const float* bias, void* out, int B, int W, int cin, int cout, int ostride,
             int ooff, hipStream_t st) {
#define SRL_CASE(CI, CO, WW) if (cin == CI && cout == CO && W == WW) return launch_gemm<CI, CO, WW, X3>(in, wfrag, bias, out, B, ostride, ooff, st);
  // 128^2 observations: the left U-Net's levels at 32^2 / 16^2 / 8^2, the right U-Net's bottom at 8^2
  SRL_CASE(32, 64, 32) SRL_CASE(64, 64, 32) SRL_CASE(128, 64, 32)
  SRL_CASE(64, 128, 16) SRL_CASE(128, 128, 16) SRL_CASE(256, 128, 16)
  SRL_CASE(128, 256, 8) SRL_CASE(256, 256, 8)
  SRL_CASE(32, 64, 8) SRL_CASE(64, 64, 8)
  // 64^2 observations (resolution_factor 4): one level further down
  SRL_CASE(32, 64, 16) SRL_CASE(64, 64, 16) SRL_CASE(128, 64, 16)
  SRL_CASE(64, 128, 8) SRL_CASE(128, 128, 8) SRL_CASE(256, 128, 8)
  SRL_CASE(128, 256, 4) SRL_CASE(256, 256, 4)
#undef SRL_CASE
  snprintf(gm_err, sizeof gm_err, "srl_conv3x3_gemm: unsupported layer %d -> %d at %d x %d", cin, cout, W, W);
  return 1;
}

}  // namespace

extern "C" {

const char* srl_conv_gemm_last_error(void) { return gm_err; }

int32_t srl_conv3x3_gemm_supported(int32_t cin, int32_t cout, int32_t W) {
  if (cout == 64) return ((W == 32 || W == 16) && (cin == 32 || cin == 64 || cin == 128)) || (W == 8 && (cin == 32 || cin == 64));
  if (cout == 128) return (W == 16 || W == 8) && (cin == 64 || cin == 128 || cin == 256);
  if (cout == 256) return (W == 8 || W == 4) && (cin == 128 || cin == 256);
  return 0;
}

// maps per workgroup: the batch must be a multiple of it
int32_t srl_conv3x3_gemm_batch_multiple(int32_t cout, int32_t W) {
  const int pxt = 128 * (4 / (cout / 64));
  return pxt >= W * W ? pxt / (W * W) : 1;
}

int64_t srl_conv3x3_gemm_wfrag_elems(int32_t cin, int32_t cout) {
  if (cin % 32 || cout % 64) return -1;
  return (int64_t)(cin / 32) * 9 * (cout / 16) * 64 * 8;
}

int srl_conv3x3_gemm_bias_relu(const void* in, const void* wfrag, const float* bias, void* out, int32_t B, int32_t W,
                               int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset, int32_t f32, void* stream) {
  if (!in || !wfrag || !bias || !out || B < 1 || !srl_conv3x3_gemm_supported(cin, cout, W) || out_stride % 4 ||
      out_offset % 4 || B % srl_conv3x3_gemm_batch_multiple(cout, W)) {
    snprintf(gm_err, sizeof gm_err, "srl_conv3x3_gemm_bias_relu: bad arguments (layers: srl_conv3x3_gemm_supported; the batch a "
             "multiple of srl_conv3x3_gemm_batch_multiple)");
    return 1;
  }
  if (f32) return dispatch<true>(in, wfrag, bias, out, B, W, cin, cout, out_stride, out_offset, (hipStream_t)stream);
  return dispatch<false>(in, wfrag, bias, out, B, W, cin, cout, out_stride, out_offset, (hipStream_t)stream);
}

int32_t srl_convt2x2_gemm_supported(int32_t cin, int32_t cout) {
  return (cin == 128 && cout == 64) || (cin == 256 && cout == 128);
}

int srl_convt2x2_gemm_bias_relu(const void* in, const void* wfrag, const float* bias, void* out, int32_t B, int32_t H, int32_t W,
                                int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset, int32_t f32, void* stream) {
  if (!in || !wfrag || !bias || !out || B < 1 || H < 1 || W < 1 || !srl_convt2x2_gemm_supported(cin, cout) || out_stride % 4 ||
      out_offset % 4) {
    snprintf(gm_err, sizeof gm_err, "srl_convt2x2_gemm_bias_relu: bad arguments (128 -> 64 or 256 -> 128 channels)");
    return 1;
  }
  const long long npix = (long long)B * H * W;
  const dim3 grid((unsigned)((npix + 127) / 128), (unsigned)(4 * cout / 256)), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define SRL_CT(CI, CO) \
  if (cin == CI && cout == CO) { \
    if (f32) hipLaunchKernelGGL((k_convt2x2_gemm<CI, CO, true>), grid, blk, 0, st, in, (const uint16_t*)wfrag, bias, out, H, W, out_stride, out_offset, npix); \
    else hipLaunchKernelGGL((k_convt2x2_gemm<CI, CO, false>), grid, blk, 0, st, in, (const uint16_t*)wfrag, bias, out, H, W, out_stride, out_offset, npix); \
  }
  SRL_CT(128, 64) SRL_CT(256, 128)
#undef SRL_CT
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(gm_err, sizeof gm_err, "srl_convt2x2_gemm_bias_relu: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

}  // extern "C"
