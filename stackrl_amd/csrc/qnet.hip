// qnet.hip — hand-written ops of the Q-network rollout path (include/stackrl_qnet.h), gfx950.
//
// K6 xcorr: out[b,u,v] = sum_{c,i,j} x[b,c,u+i,v+j] * w[b,c,i,j]   (layers.py:21-38)
//   fp32 FMA on the vector ALU (the reference computes in fp32; gfx950 has no faster exact-fp32 matrix path:
//   v_mfma_f32_*_f32 runs at the vector rate).  One thread owns a strip of 8 adjacent outputs of one row: for
//   every (channel, kernel row) it loads the 8 + kw - 1 inputs the strip needs into registers once and sweeps the
//   kernel row over them; the kernel row w[b,c,i,:] is wave-uniform, so it comes in through scalar loads and sits
//   in SGPRs as the FMA operand — no LDS traffic at all, 256 FMAs per 39 vector loads.
//
// policy head: one workgroup per env row; (value, index) arg-max with lowest-index ties, then epsilon-greedy select.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stackrl_qnet.h"

namespace {
thread_local char q_err[256] = "";

template <int KW, int SW>
__global__ void __launch_bounds__(256) k_xcorr_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   float* __restrict__ out, int C, int H, int W, int kh, int OH, int OW) {
  const int b = blockIdx.y;
  const int u = blockIdx.x * blockDim.y + threadIdx.y;
  const int v0 = threadIdx.x * SW;
  const bool active = (u < OH) && (v0 < OW);
  const int uc = u < OH ? u : OH - 1;
  float acc[SW];
#pragma unroll
  for (int k = 0; k < SW; ++k) acc[k] = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float* xc = x + ((size_t)b * C + c) * H * W;
    const float* wc = w + ((size_t)b * C + c) * kh * KW;     // uniform across the wave: scalar loads
    for (int i = 0; i < kh; ++i) {
      const float* xr = xc + (size_t)(uc + i) * W;
      float xs[SW + KW - 1];
#pragma unroll
      for (int t = 0; t < SW + KW - 1; ++t) {
        int col = v0 + t;
        xs[t] = xr[col < W ? col : W - 1];                     // clamped: only feeds outputs that are not stored
      }
      const float* wr = wc + i * KW;
#pragma unroll
      for (int j = 0; j < KW; ++j) {
        const float wj = wr[j];
#pragma unroll
        for (int k = 0; k < SW; ++k) acc[k] = fmaf(xs[k + j], wj, acc[k]);
      }
    }
  }
  if (active) {
    float* o = out + ((size_t)b * OH + u) * OW + v0;
#pragma unroll
    for (int k = 0; k < SW; ++k)
      if (v0 + k < OW) o[k] = acc[k];
  }
}

__global__ void __launch_bounds__(256) k_policy_head(const float* __restrict__ adv, const float* __restrict__ u,
                                                     const int64_t* __restrict__ rnd, float eps,
                                                     int64_t* __restrict__ actions, int A) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* a = adv + (size_t)b * A;
  float best = -3.0e38f; int bi = 0x7fffffff;
  for (int k = tid; k < A; k += 256) {
    float t = a[k];
    if (t > best) { best = t; bi = k; }          // ascending k per thread: lowest index of its maxima
  }
  sv[tid] = best; si[tid] = bi;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (tid < s) {
      float ov = sv[tid + s]; int oi = si[tid + s];
      if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
    }
    __syncthreads();
  }
  if (tid == 0) actions[b] = (u[b] > eps) ? (int64_t)si[0] : rnd[b];   // tf.where(uniform > e, argmax, random), dqn.py:336-348
}
}  // namespace

extern "C" {

const char* srl_qnet_last_error(void) { return q_err; }

#ifndef SRL_BUILD_INFO
#define SRL_BUILD_INFO "SRL_BUILD_INFO<unknown|>"
#endif
const char* srl_qnet_build_info(void) { static const char info[] = SRL_BUILD_INFO; return info; }

int srl_xcorr_forward(const float* x, const float* w, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                      int32_t kh, int32_t kw, void* stream) {
  if (!x || !w || !out || B < 1 || C < 1 || kh < 1 || H < kh || W < kw) {
    snprintf(q_err, sizeof q_err, "srl_xcorr_forward: bad arguments");
    return 1;
  }
  const int OH = H - kh + 1, OW = W - kw + 1;
  constexpr int SW = 8;
  const int strips = (OW + SW - 1) / SW;
  if (strips > 64) { snprintf(q_err, sizeof q_err, "srl_xcorr_forward: output row too wide"); return 1; }
  const int rows = 256 / strips > 0 ? 256 / strips : 1;
  dim3 block(strips, rows), grid((OH + rows - 1) / rows, B);
  hipStream_t st = (hipStream_t)stream;
  if (kw == 32) hipLaunchKernelGGL((k_xcorr_fwd<32, SW>), grid, block, 0, st, x, w, out, C, H, W, kh, OH, OW);
  else if (kw == 16) hipLaunchKernelGGL((k_xcorr_fwd<16, SW>), grid, block, 0, st, x, w, out, C, H, W, kh, OH, OW);
  else { snprintf(q_err, sizeof q_err, "srl_xcorr_forward: kw must be 16 or 32"); return 1; }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(q_err, sizeof q_err, "srl_xcorr_forward: %s", hipGetErrorString(e)); return 4; }
  return 0;
}

int srl_policy_head(const float* adv, const float* u, const int64_t* rnd, float epsilon, int64_t* actions, int32_t B,
                    int32_t A, void* stream) {
  if (!adv || !u || !rnd || !actions || B < 1 || A < 1) {
    snprintf(q_err, sizeof q_err, "srl_policy_head: bad arguments");
    return 1;
  }
  hipLaunchKernelGGL(k_policy_head, dim3(B), dim3(256), 0, (hipStream_t)stream, adv, u, rnd, epsilon, actions, A);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(q_err, sizeof q_err, "srl_policy_head: %s", hipGetErrorString(e)); return 4; }
  return 0;
}

}  // extern "C"
