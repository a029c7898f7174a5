// heuristics.hip — the reference's heuristic baseline policies as device kernels (float64 like numpy).
//
// stackrl/baselines.py: `height` :28-43, `difference` :45-77, `corrcoef` :79-114, `correlate` :141-143,
// `goal_overlap` :152-156 and the selection of `Baseline.call` :201-217.  All of them are sliding-window reductions of
// the (h x w) object map over the (H x W) height map — (H-h+1)(W-w+1) windows of h*w taps — which the reference runs
// as double Python loops per observation.  Here: one 256-thread workgroup per env, the uint8 maps staged once in LDS
// (16 KB + 1 KB), a 256-entry table k -> k / gmax so that every tap value is the same double numpy computes, one
// thread per window (strided), float64 accumulation in row-major tap order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stackrl_qnet.h"

namespace {
thread_local char h_err[256] = "";
#define q_err h_err



enum { M_CORRELATE = 1, M_HEIGHT = 2, M_DIFFERENCE = 3, M_CORRCOEF = 4 };

__device__ __forceinline__ double ipow(double x, int e) {   // numpy: x**1 = x, x**2 = x*x; otherwise pow
  if (e == 1) return x;
  if (e == 2) return x * x;
  return pow(x, (double)e);
}

__global__ void __launch_bounds__(256)
k_heuristic(int method, const uint8_t* __restrict__ obs_map, const uint8_t* __restrict__ obs_obj,
            double* __restrict__ values, uint8_t* __restrict__ mask, int H, int h, int dexp, int wexp, int localized,
            double threshold) {
  extern __shared__ unsigned char lds[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int OH = H - h + 1, A = OH * OH;
  uint8_t* sm = lds;                    // [H*H] height channel
  uint8_t* sg = sm + H * H;             // [H*H] (height < goal) flags
  uint8_t* so = sg + H * H;             // [h*h] object map
  double* lut = (double*)(so + ((h * h + 7) & ~7));   // [256] k / gmax
  double* wts = lut + 256;              // [h*h] difference weights / centred n for corrcoef
  double* red = wts + h * h;            // [256] reductions
  int* redi = (int*)(red + 256);        // [256]
  const uint8_t* m = obs_map + (size_t)b * H * H * 2;
  const uint8_t* o = obs_obj + (size_t)b * h * h;
  int gl = 0;
  for (int k = tid; k < H * H; k += 256) {
    uint8_t hv = m[2 * k], gv = m[2 * k + 1];
    sm[k] = hv; sg[k] = hv < gv;
    gl = gl > gv ? gl : gv;
  }
  for (int k = tid; k < h * h; k += 256) so[k] = o[k];
  redi[tid] = gl;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (tid < s) redi[tid] = redi[tid] > redi[tid + s] ? redi[tid] : redi[tid + s]; __syncthreads(); }
  const double gmax = (double)redi[0];                    // baselines.py:23
  __syncthreads();
  lut[tid] = (double)tid / gmax;
  __syncthreads();
  // ---- per-method constants of the object map
  double nsum = 0.0, ncount = 0.0, nmean = 0.0, nvar = 0.0, wsum = 0.0;
  if (method == M_DIFFERENCE) {
    double part = 0.0;
    for (int k = tid; k < h * h; k += 256) {
      int i = k / h, j = k - i * h;
      double w = 0.0;
      if (so[k] > 0) {
        if (wexp > 0) {
          double wi = ((double)i - h / 2.0), wj = ((double)j - h / 2.0);
          double r2 = wi * wi + wj * wj;
          w = wexp == 2 ? r2 : pow(r2, wexp / 2.0);          // (wi^2 + wj^2) ** (wexp / 2)
        } else w = 1.0;
      }
      wts[k] = w; part += w;
    }
    red[tid] = part; __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    wsum = red[0]; __syncthreads();
    for (int k = tid; k < h * h; k += 256) wts[k] = wts[k] / wsum;
    __syncthreads();
  } else if (method == M_CORRCOEF || method == M_CORRELATE) {
    double part = 0.0, cnt = 0.0;
    for (int k = tid; k < h * h; k += 256) {
      bool in = localized ? so[k] > 0 : true;
      if (method == M_CORRELATE) in = true;
      if (in) { part += lut[so[k]]; cnt += 1.0; }
    }
    red[tid] = part; __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    nsum = red[0]; __syncthreads();
    red[tid] = cnt; __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    ncount = red[0]; __syncthreads();
    if (method == M_CORRCOEF) {
      nmean = nsum / ncount;
      double pv = 0.0;
      for (int k = tid; k < h * h; k += 256) {
        bool in = localized ? so[k] > 0 : true;
        double c = lut[so[k]] - nmean;                       // n -= mean (baselines.py:94)
        wts[k] = c;
        if (in) pv += c * c;
      }
      red[tid] = pv; __syncthreads();
      for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
      nvar = red[0]; __syncthreads();
    }
  }
  // ---- windows
  int fmax_local = 0;
  for (int a = tid; a < A; a += 256) {
    const int i = a / OH, j = a - i * OH;
    double f = 0.0;
    if (method == M_HEIGHT || method == M_DIFFERENCE) {
      double h0 = 0.0;                                       // np.max(np.where(n_where, o + n, 0))
      for (int p = 0; p < h; ++p)
        for (int q = 0; q < h; ++q) {
          uint8_t nv = so[p * h + q];
          if (nv > 0) { double s = lut[sm[(i + p) * H + j + q]] + lut[nv]; h0 = s > h0 ? s : h0; }
        }
      if (method == M_HEIGHT) f = h0;
      else {
        for (int p = 0; p < h; ++p)
          for (int q = 0; q < h; ++q) {
            double hh = lut[sm[(i + p) * H + j + q]] + lut[so[p * h + q]];
            f += wts[p * h + q] * ipow(fabs(h0 - hh), dexp);
          }
      }
    } else if (method == M_CORRELATE) {
      for (int p = 0; p < h; ++p)
        for (int q = 0; q < h; ++q) f += lut[sm[(i + p) * H + j + q]] * lut[so[p * h + q]];
      f = f / nsum;
    } else if (method == M_CORRCOEF) {
      if (nvar != 0.0) {
        double osum = 0.0;
        for (int p = 0; p < h; ++p)
          for (int q = 0; q < h; ++q)
            if (!localized || so[p * h + q] > 0) osum += lut[sm[(i + p) * H + j + q]];
        const double omean = osum / ncount;
        double ovar = 0.0, num = 0.0;
        for (int p = 0; p < h; ++p)
          for (int q = 0; q < h; ++q)
            if (!localized || so[p * h + q] > 0) {
              double c = lut[sm[(i + p) * H + j + q]] - omean;
              ovar += c * c; num += wts[p * h + q] * c;
            }
        if (ovar != 0.0) f = num / sqrt(nvar * ovar);
      }
    }
    values[(size_t)b * A + a] = f;
    if (mask) {                                              // goal_overlap numerator (integers): running maximum
      int ov = 0;
      for (int p = 0; p < h; ++p)
        for (int q = 0; q < h; ++q) ov += (so[p * h + q] > 0) & sg[(i + p) * H + j + q];
      fmax_local = fmax_local > ov ? fmax_local : ov;
    }
  }
  if (!mask) return;
  __syncthreads();
  redi[tid] = fmax_local; __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (tid < s) redi[tid] = redi[tid] > redi[tid + s] ? redi[tid] : redi[tid + s]; __syncthreads(); }
  const double thr = threshold * (double)redi[0];            // f >= threshold * f.max()  (baselines.py:156)
  for (int a = tid; a < A; a += 256) {
    const int i = a / OH, j = a - i * OH;
    int ov = 0;
    for (int p = 0; p < h; ++p)
      for (int q = 0; q < h; ++q) ov += (so[p * h + q] > 0) & sg[(i + p) * H + j + q];
    mask[(size_t)b * A + a] = (double)ov >= thr ? 1 : 0;
  }
}

// Baseline.call (baselines.py:201-217): goal mask, optional local-minimum filter (scipy minimum_filter, size
// 1 + 2 minorder, mode 'constant' -> 0 outside), arg-min with first-occurrence ties; also the negated value map.
__global__ void __launch_bounds__(256)
k_baseline_select(const double* __restrict__ values, const uint8_t* __restrict__ mask, int use_goal, int minorder,
                  int64_t* __restrict__ actions, double* __restrict__ neg_values, int OH) {
  __shared__ double sv[256];
  __shared__ int si[256];
  __shared__ int any_minima;
  const int b = blockIdx.x, tid = threadIdx.x, A = OH * OH;
  const double* v = values + (size_t)b * A;
  const uint8_t* mk = mask + (size_t)b * A;
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  if (tid == 0) any_minima = 0;
  __syncthreads();
  // pass 1: is there any masked local minimum?  and max of masked values (for the returned map)
  double vmax = -INF; int found = 0;
  for (int a = tid; a < A; a += 256) {
    if (use_goal && mk[a]) {
      vmax = v[a] > vmax ? v[a] : vmax;
      if (minorder > 0) {
        const int i = a / OH, j = a - i * OH;
        double mn = INF;
        for (int di = -minorder; di <= minorder; ++di)
          for (int dj = -minorder; dj <= minorder; ++dj) {
            const int ii = i + di, jj = j + dj;
            const double t = (ii >= 0 && ii < OH && jj >= 0 && jj < OH) ? v[ii * OH + jj] : 0.0;
            mn = t < mn ? t : mn;
          }
        if (mn == v[a]) found = 1;
      }
    }
  }
  if (found) atomicOr(&any_minima, 1);
  sv[tid] = vmax; __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (tid < s) sv[tid] = sv[tid] > sv[tid + s] ? sv[tid] : sv[tid + s]; __syncthreads(); }
  const double masked_max = sv[0];
  const bool use_minima = use_goal && minorder > 0 && any_minima;
  __syncthreads();
  // pass 2: arg-min over the candidate set
  double best = INF; int bi = 0x7fffffff;
  for (int a = tid; a < A; a += 256) {
    bool cand = true;
    if (use_goal) {
      cand = mk[a];
      if (cand && use_minima) {
        const int i = a / OH, j = a - i * OH;
        double mn = INF;
        for (int di = -minorder; di <= minorder; ++di)
          for (int dj = -minorder; dj <= minorder; ++dj) {
            const int ii = i + di, jj = j + dj;
            const double t = (ii >= 0 && ii < OH && jj >= 0 && jj < OH) ? v[ii * OH + jj] : 0.0;
            mn = t < mn ? t : mn;
          }
        cand = (mn == v[a]);
      }
    }
    const double t = cand ? v[a] : INF;
    if (t < best || (t == best && a < bi)) { best = t; bi = a; }
    if (neg_values) neg_values[(size_t)b * A + a] = use_goal ? -(mk[a] ? v[a] : masked_max + 0.001) : -v[a];
  }
  sv[tid] = best; si[tid] = bi; __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (tid < s && (sv[tid + s] < sv[tid] || (sv[tid + s] == sv[tid] && si[tid + s] < si[tid]))) { sv[tid] = sv[tid + s]; si[tid] = si[tid + s]; }
    __syncthreads();
  }
  if (tid == 0) actions[b] = (int64_t)(si[0] == 0x7fffffff ? 0 : si[0]);   // np.argmin of an all-inf array is 0
}

}  // namespace

extern "C" {

int srl_heuristic(int32_t method, const uint8_t* obs_map, const uint8_t* obs_obj, double* values, uint8_t* mask,
                  int32_t B, int32_t H, int32_t h, int32_t difference_exponent, int32_t weights_exponent,
                  int32_t localized, double threshold, void* stream) {
  if (!obs_map || !obs_obj || !values || B < 1 || h < 1 || H < h || method < 1 || method > 4) {
    snprintf(q_err, sizeof q_err, "srl_heuristic: bad arguments");
    return 1;
  }
  size_t lds = 2 * (size_t)H * H + (((size_t)h * h + 7) & ~(size_t)7) + sizeof(double) * (256 + (size_t)h * h + 256) + sizeof(int) * 256;
  hipFuncSetAttribute((const void*)k_heuristic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_heuristic, dim3(B), dim3(256), lds, (hipStream_t)stream, method, obs_map, obs_obj, values, mask,
                     H, h, difference_exponent, weights_exponent, localized, threshold);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(q_err, sizeof q_err, "srl_heuristic: %s", hipGetErrorString(e)); return 4; }
  return 0;
}

int srl_baseline_select(const double* values, const uint8_t* mask, int32_t use_goal, int32_t minorder,
                        int64_t* actions, double* neg_values, int32_t B, int32_t OH, void* stream) {
  if (!values || !actions || (use_goal && !mask) || B < 1 || OH < 1 || minorder < 0) {
    snprintf(q_err, sizeof q_err, "srl_baseline_select: bad arguments");
    return 1;
  }
  hipLaunchKernelGGL(k_baseline_select, dim3(B), dim3(256), 0, (hipStream_t)stream, values, mask ? mask : (const uint8_t*)values,
                     use_goal, minorder, actions, neg_values, OH);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(q_err, sizeof q_err, "srl_baseline_select: %s", hipGetErrorString(e)); return 4; }
  return 0;
}

}  // extern "C"
