// epilogue.hip — fused element-wise passes of the Q-net rollout forward (inference only), gfx950.
//
// The U-Net convolutions are library calls (MIOpen); around every one of them the stock graph runs a bias add, a
// ReLU, and — per level — a 2 x 2 max-pool, a concatenation copy and finally a layout change for the
// cross-correlation: 45 % of the rollout's GPU time in separate memory-bound kernels.  These two kernels do that work
// in one pass over each convolution output (channels-last, `[pixel][channel]` in memory):
//
//   k_bias_act       y = relu(x + bias[c])  written in place, or into a channel slice of a wider channels-last
//                    buffer (the decoder's concatenation buffer: no torch.cat), or transposed to [channel][pixel]
//                    (the layout the cross-correlation kernel stages from)
//   k_bias_act_pool  the same, plus the 2 x 2 max-pooled copy for the next encoder level
//
// Reference ops replaced: Conv2D bias + activation='relu', MaxPool2D, Concatenate of `layers.unet`
// (stackrl/nets/layers.py:135-259).  8 channels (16 bytes of bf16) per thread.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stackrl_qnet.h"
#include "srl_bf16.h"

namespace {

__device__ __forceinline__ uint32_t e_bf16_rne(float f) { return srl_bf16(f); }
__device__ __forceinline__ void unpack8(const uint4 q, float* v) {
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint32_t w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = srl_pk_bf16(v[2 * k], v[2 * k + 1]);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// element types: bf16 (uint16_t, 16 bytes per 8 channels) or float (the fp32 rollout of the reference's dtype, 32 bytes)
__device__ __forceinline__ void load8(const uint16_t* p, float* v) { unpack8(*(const uint4*)p, v); }
__device__ __forceinline__ void load8(const float* p, float* v) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(uint16_t* p, const float* v) { *(uint4*)p = pack8(v); }
__device__ __forceinline__ void store8(float* p, const float* v) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store1(uint16_t* p, float v) { *p = (uint16_t)e_bf16_rne(v); }
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ float stored(const uint16_t*, float v) { return __uint_as_float(e_bf16_rne(v) << 16); }   // the value as stored
__device__ __forceinline__ float stored(const float*, float v) { return v; }

// in [npix][C]; out: channels-last slice (pixel stride ostride, channel offset ooff) or, when nchw_hw > 0,
// [B][C][nchw_hw] with npix = B * nchw_hw
template <typename E>
__global__ void __launch_bounds__(256) k_bias_act(const E* __restrict__ in, E* __restrict__ out,
                                                  const float* __restrict__ bias, long long npix, int C, int ostride,
                                                  int ooff, int nchw_hw, int relu) {
  const int cg = C / 8;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npix * cg) return;
  long long pix; int g;
  if (nchw_hw > 0) { g = (int)(idx / npix); pix = idx - (long long)g * npix; }   // adjacent lanes = adjacent pixels
  else { pix = idx / cg; g = (int)(idx - pix * cg); }
  float v[8];
  load8(in + pix * C + g * 8, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] += bias[g * 8 + k];
    if (relu) v[k] = fmaxf(v[k], 0.0f);
  }
  if (nchw_hw > 0) {
    const long long b = pix / nchw_hw, p = pix - b * nchw_hw;
    E* o = out + (b * C + g * 8) * nchw_hw + p;
#pragma unroll
    for (int k = 0; k < 8; ++k) store1(o + (long long)k * nchw_hw, v[k]);
  } else {
    store8(out + pix * ostride + ooff + g * 8, v);
  }
}

// in [B][H][W][C] (H, W even); skip out: channels-last slice as above; pooled [B][H/2][W/2][C]
template <typename E>
__global__ void __launch_bounds__(256) k_bias_act_pool(const E* __restrict__ in, E* __restrict__ skip,
                                                       E* __restrict__ pooled, const float* __restrict__ bias,
                                                       int B, int H, int W, int C, int ostride, int ooff) {
  const int cg = C / 8, H2 = H / 2, W2 = W / 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)B * H2 * W2 * cg) return;
  const int g = (int)(idx % cg);
  long long q = idx / cg;
  const int x2 = (int)(q % W2); q /= W2;
  const int y2 = (int)(q % H2);
  const long long b = q / H2;
  float bz[8], m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { bz[k] = bias[g * 8 + k]; m[k] = 0.0f; }   // post-ReLU values are >= 0
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const long long pix = (b * H + 2 * y2 + dy) * W + 2 * x2 + dx;
      float v[8];
      load8(in + pix * C + g * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] + bz[k], 0.0f);
      store8(skip + pix * ostride + ooff + g * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], stored(in, v[k]));   // pool the values as stored, like MaxPool2D on the tensor
    }
  store8(pooled + ((b * H2 + y2) * W2 + x2) * C + g * 8, m);
}


// 2 x 2 max-pool of a finished activation (channels-last, possibly a channel slice of a wider buffer: pixel stride
// istride, channel offset ioff) into a contiguous channels-last tensor [B][H/2][W/2][C]; 8 channels per thread
template <typename E>
__global__ void __launch_bounds__(256) k_pool2x2(const E* __restrict__ in, E* __restrict__ pooled, int B, int H, int W, int C,
                                                 int istride, int ioff) {
  const int cg = C / 8, H2 = H / 2, W2 = W / 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)B * H2 * W2 * cg) return;
  const int g = (int)(idx % cg);
  long long q = idx / cg;
  const int x2 = (int)(q % W2); q /= W2;
  const int y2 = (int)(q % H2);
  const long long b = q / H2;
  float m[8];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const long long pix = (b * H + 2 * y2 + dy) * W + 2 * x2 + dx;
      float v[8];
      load8(in + pix * istride + ioff + g * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = (dy | dx) ? fmaxf(m[k], v[k]) : v[k];
    }
  store8(pooled + ((b * H2 + y2) * W2 + x2) * C + g * 8, m);
}

// Backward of y = relu(x + bias) (or y = x + bias) for the update path (`DQN.train`, agents/dqn.py:466-469): the gradient
// with respect to x, gx = gy * (y > 0), and the per-block partial sums of the bias gradient, one pass over gy and y.
// float32 channels-last [npix][C]; a block walks `pixb` consecutive pixels, thread = (pixel lane, 8-channel group).
// The partials are summed in index order by k_bias_grad_finish: a fixed summation order and no cross-block
// synchronisation inside a kernel.
__global__ void __launch_bounds__(256) k_bias_act_bwd(const float* __restrict__ gy, const float* __restrict__ y,
                                                      float* __restrict__ gx, float* __restrict__ partial, long long npix,
                                                      int C, int relu, int pixb) {
  __shared__ float sh[256 * 8];
  const int cg = C / 8, ppi = 256 / cg;          // channel groups, pixels per iteration
  const int g = threadIdx.x % cg, pl = threadIdx.x / cg;
  const long long p0 = (long long)blockIdx.x * pixb;
  long long p1 = p0 + pixb; if (p1 > npix) p1 = npix;
  float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (pl < ppi)
    for (long long p = p0 + pl; p < p1; p += ppi) {
      float a[8], b[8];
      load8(gy + p * C + g * 8, a);
      if (relu) {
        load8(y + p * C + g * 8, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = b[k] > 0.0f ? a[k] : 0.0f;
      }
      store8(gx + p * C + g * 8, a);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += a[k];
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[threadIdx.x * 8 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < C) {   // channel c: the partial sums of its group's pixel lanes, in lane order
    const int c = threadIdx.x, gg = c / 8, k = c % 8;
    float s = 0.0f;
    for (int q = 0; q < ppi; ++q) s += sh[(q * cg + gg) * 8 + k];
    partial[(long long)blockIdx.x * C + c] = s;
  }
}

// gb[c] = sum over blocks of partial[b][c], in a fixed order: a workgroup takes 32 channels (or all C < 32), its 256
// threads are (row r, channel): row r adds the blocks r, r + rows, ... in turn (coalesced 128-byte reads), then the rows
// are added in index order.
__global__ void __launch_bounds__(256) k_bias_grad_finish(const float* __restrict__ partial, int nblk, int C,
                                                          float* __restrict__ gb) {
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

// pixels per block of k_bias_act_bwd: at most 1,024 blocks, whole iterations of the block's pixel lanes
static int bwd_pixb(long long npix, int C) {
  const int ppi = 256 / (C / 8);
  long long pixb = (npix + 1023) / 1024;
  if (pixb < 4 * ppi) pixb = 4 * ppi;
  return (int)((pixb + ppi - 1) / ppi * ppi);
}

thread_local char e_err[256] = "";

int finish(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(e_err, sizeof e_err, "%s: %s", what, hipGetErrorString(e)); return 2; }
  return 0;
}

}  // namespace

extern "C" {

const char* srl_epilogue_last_error(void) { return e_err; }

int srl_bias_act(const void* in, void* out, const float* bias, int64_t npix, int32_t C, int32_t out_stride,
                 int32_t out_offset, int32_t nchw_hw, int32_t relu, void* stream) {
  if (!in || !out || !bias || npix < 1 || C < 8 || C % 8 || out_stride % 8 || out_offset % 8 ||
      (nchw_hw > 0 && npix % nchw_hw)) {
    snprintf(e_err, sizeof e_err, "srl_bias_act: bad arguments (channel counts, strides and offsets must be multiples of 8)");
    return 1;
  }
  const long long n = npix * (C / 8);
  hipLaunchKernelGGL(k_bias_act<uint16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in,
                     (uint16_t*)out, bias, (long long)npix, C, out_stride, out_offset, nchw_hw, relu);
  return finish("srl_bias_act");
}

int srl_bias_act_f32(const float* in, float* out, const float* bias, int64_t npix, int32_t C, int32_t out_stride,
                     int32_t out_offset, int32_t nchw_hw, int32_t relu, void* stream) {
  if (!in || !out || !bias || npix < 1 || C < 8 || C % 8 || out_stride % 8 || out_offset % 8 ||
      (nchw_hw > 0 && npix % nchw_hw)) {
    snprintf(e_err, sizeof e_err, "srl_bias_act_f32: bad arguments (channel counts, strides and offsets must be multiples of 8)");
    return 1;
  }
  const long long n = npix * (C / 8);
  hipLaunchKernelGGL(k_bias_act<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, bias,
                     (long long)npix, C, out_stride, out_offset, nchw_hw, relu);
  return finish("srl_bias_act_f32");
}

int srl_bias_act_pool(const void* in, void* skip, void* pooled, const float* bias, int32_t B, int32_t H, int32_t W,
                      int32_t C, int32_t skip_stride, int32_t skip_offset, void* stream) {
  if (!in || !skip || !pooled || !bias || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || C % 8 ||
      skip_stride % 8 || skip_offset % 8) {
    snprintf(e_err, sizeof e_err, "srl_bias_act_pool: bad arguments");
    return 1;
  }
  const long long n = (long long)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(k_bias_act_pool<uint16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)in, (uint16_t*)skip, (uint16_t*)pooled, bias, B, H, W, C, skip_stride, skip_offset);
  return finish("srl_bias_act_pool");
}

int srl_bias_act_pool_f32(const float* in, float* skip, float* pooled, const float* bias, int32_t B, int32_t H, int32_t W,
                          int32_t C, int32_t skip_stride, int32_t skip_offset, void* stream) {
  if (!in || !skip || !pooled || !bias || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || C % 8 ||
      skip_stride % 8 || skip_offset % 8) {
    snprintf(e_err, sizeof e_err, "srl_bias_act_pool_f32: bad arguments");
    return 1;
  }
  const long long n = (long long)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(k_bias_act_pool<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, skip,
                     pooled, bias, B, H, W, C, skip_stride, skip_offset);
  return finish("srl_bias_act_pool_f32");
}

int64_t srl_bias_act_bwd_scratch_floats(int64_t npix, int32_t C) {
  if (npix < 1 || C < 8 || C % 8 || C > 256 || 256 % (C / 8)) return -1;
  const int pixb = bwd_pixb(npix, C);
  return ((npix + pixb - 1) / pixb) * C;
}

int srl_bias_act_bwd_f32(const float* gy, const float* y, float* gx, float* gbias, float* scratch, int64_t npix, int32_t C,
                         int32_t relu, void* stream) {
  if (!gy || (relu && !y) || !gx || !gbias || !scratch || srl_bias_act_bwd_scratch_floats(npix, C) < 0) {
    snprintf(e_err, sizeof e_err, "srl_bias_act_bwd_f32: bad arguments (C a multiple of 8 that divides 2,048, at most 256)");
    return 1;
  }
  const int pixb = bwd_pixb(npix, C);
  const int nblk = (int)((npix + pixb - 1) / pixb);
  hipLaunchKernelGGL(k_bias_act_bwd, dim3(nblk), dim3(256), 0, (hipStream_t)stream, gy, y, gx, scratch, (long long)npix, C,
                     relu, pixb);
  hipLaunchKernelGGL(k_bias_grad_finish, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, scratch, nblk, C, gbias);
  return finish("srl_bias_act_bwd_f32");
}

int srl_pool2x2(const void* in, void* pooled, int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_stride, int32_t in_offset,
                int32_t f32, void* stream) {
  if (!in || !pooled || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || C % 8 || in_stride % 8 || in_offset % 8) {
    snprintf(e_err, sizeof e_err, "srl_pool2x2: bad arguments");
    return 1;
  }
  const long long n = (long long)B * (H / 2) * (W / 2) * (C / 8);
  const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
  if (f32) hipLaunchKernelGGL(k_pool2x2<float>, grid, blk, 0, (hipStream_t)stream, (const float*)in, (float*)pooled, B, H, W, C, in_stride, in_offset);
  else hipLaunchKernelGGL(k_pool2x2<uint16_t>, grid, blk, 0, (hipStream_t)stream, (const uint16_t*)in, (uint16_t*)pooled, B, H, W, C, in_stride, in_offset);
  return finish("srl_pool2x2");
}

}  // extern "C"
