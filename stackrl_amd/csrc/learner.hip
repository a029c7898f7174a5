// learner.hip — hand-written kernels of the DQN update path (libstackrl_qnet.so, include/stackrl_qnet.h):
//
//   k_td_epilogue      Double-DQN target, TD error, Huber loss with importance weights, the new replay priorities and
//                      the gradient of the loss with respect to Q(s, .) in one pass            (agents/dqn.py:408-476)
//   k_adam_tick/step   Keras Adam over one flat parameter bucket                                (agents/dqn.py:473)
//   k_gumbel_topk*     K7: prioritised sampling without replacement, top-k of alpha logit + Gumbel (agents/memory.py:220-223)
//   k_replay_scatter   K8: one transition per env into the replay partitions                    (agents/memory.py:153-161)
//   k_replay_gather    K8: the sampled minibatch (state, next state, action, reward, terminal, IS weight)
//                                                                                               (agents/memory.py:232-260)
// All of them are HBM- or latency-bound glue between the network passes: each replaces 10 - 30 library launches of a few
// microseconds by one launch, and everything they touch stays at fixed addresses so the update can be replayed from a
// hipGraph.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/stackrl_qnet.h"

namespace {

thread_local char g_err[256] = "";
int fail(const char* m) { snprintf(g_err, sizeof g_err, "%s", m); return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(hipGetErrorString(e_)); } while (0)

// order-preserving map float -> uint32 (larger float = larger uint)
__device__ __forceinline__ uint32_t f2o(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
// (value, lowest index) arg-max key: larger value first, then the smaller index
__device__ __forceinline__ uint64_t vkey(float v, uint32_t idx) { return ((uint64_t)f2o(v) << 32) | (uint32_t)(~idx); }

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t k) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)k, s), hi = __shfl_xor((uint32_t)(k >> 32), s);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    k = o > k ? o : k;
  }
  return k;
}

// block-wide maximum of a 64-bit key (T threads, T / 64 <= 16 waves); every thread gets the result
template <int T>
__device__ __forceinline__ uint64_t block_max_u64(uint64_t k, uint64_t* sh) {
  k = wave_max_u64(k);
  const int w = threadIdx.x >> 6;
  __syncthreads();                       // sh may still be read from the previous call
  if ((threadIdx.x & 63) == 0) sh[w] = k;
  __syncthreads();
  uint64_t r = sh[0];
#pragma unroll
  for (int i = 1; i < T / 64; ++i) r = sh[i] > r ? sh[i] : r;
  return r;
}

// ------------------------------------------------------------------------------------------------ TD epilogue
// One workgroup per sample.  ticket: a device word, zero before the launch; the workgroup that arrives last reduces the
// per-sample terms in index order (a fixed summation order) and resets the ticket, so the launch can be replayed.
template <int T>
__global__ void __launch_bounds__(T) k_td_epilogue(const float* __restrict__ q, const float* __restrict__ qn_online,
    const float* __restrict__ qn_target, const int64_t* __restrict__ actions, const float* __restrict__ rewards,
    const uint8_t* __restrict__ terminal, const float* __restrict__ weights, float gamma, float huber_delta,
    float reward_scale, int use_double, float prio_eps, int mb, int A, float* __restrict__ out_loss,
    float* __restrict__ out_mtd, float* __restrict__ td_abs, float* __restrict__ logits, float* __restrict__ grad_q,
    float* __restrict__ scratch /* [2 mb] */, int* __restrict__ ticket) {
  __shared__ uint64_t sh[T / 64];
  __shared__ int s_last;
  const int b = blockIdx.x, tid = threadIdx.x;
  // dqn.py:424-431: Double-DQN takes the target net's value at the online net's arg-max; plain DQN the target's max
  const float* sel = (use_double ? qn_online : qn_target) + (size_t)b * A;
  uint64_t best = 0;
  for (int a = tid; a < A; a += T) { const uint64_t k = vkey(sel[a], (uint32_t)a); best = k > best ? k : best; }
  best = block_max_u64<T>(best, sh);
  const int astar = (int)~(uint32_t)best;
  if (grad_q) {
    float* g = grad_q + (size_t)b * A;
    for (int a = tid; a < A; a += T) g[a] = 0.0f;
  }
  __syncthreads();
  if (tid == 0) {
    const float tq = qn_target[(size_t)b * A + astar];
    float r = rewards[b];
    if (reward_scale != 0.0f) r = r * reward_scale;
    const float y = r + (terminal[b] ? 0.0f : gamma * tq);                   // dqn.py:450-454
    const int act = (int)actions[b];
    const float td = q[(size_t)b * A + act] - y;                             // one_hot . reduce_sum, dqn.py:410-417
    const float ad = fabsf(td);
    float loss, dq;
    if (huber_delta >= 0.0f) {                                               // dqn.py:461-464
      const float quad = fminf(ad, huber_delta), lin = ad - quad;
      loss = 0.5f * quad * quad + huber_delta * lin;
      dq = td < 0.0f ? -quad : quad;
    } else {
      loss = 0.5f * ad * ad;
      dq = td;
    }
    const float w = weights ? weights[b] : 1.0f;
    loss = loss * w;
    td_abs[b] = ad;
    logits[b] = logf(ad + prio_eps);                                         // memory.py:272
    if (grad_q) grad_q[(size_t)b * A + act] = (dq * w) / (float)mb;
    scratch[b] = loss; scratch[mb + b] = td;
    __threadfence();
    s_last = atomicAdd(ticket, 1) == mb - 1;
  }
  __syncthreads();
  if (s_last && tid == 0) {
    __threadfence();
    float sl = 0.0f, st = 0.0f;
    for (int i = 0; i < mb; ++i) { sl += ((volatile float*)scratch)[i]; st += ((volatile float*)scratch)[mb + i]; }
    out_loss[0] = sl / (float)mb;
    out_mtd[0] = st / (float)mb;
    *ticket = 0;
  }
}

// ------------------------------------------------------------------------------------------------ Adam (Keras)
// state: [0] step t (as float), [1] beta1^t, [2] beta2^t, [3] lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t)
__global__ void k_adam_tick(float* __restrict__ state, float lr, float b1, float b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float t = state[0] + 1.0f;
    const float p1 = state[1] * b1, p2 = state[2] * b2;
    state[0] = t; state[1] = p1; state[2] = p2;
    state[3] = lr * sqrtf(1.0f - p2) / (1.0f - p1);
  }
}
// Keras `Adam._resource_apply_dense` (epsilon outside the bias correction): m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2);
// p -= lr_t m / (sqrt(v) + eps).  16 bytes per lane per stream; n4 = n / 4 vector groups, the tail is handled scalar.
__global__ void __launch_bounds__(256) k_adam_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, const float* __restrict__ state,
                                                    float b1, float b2, float eps) {
  const float lrt = state[3];
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n4 = n >> 2;
  if (i4 < n4) {
    float4 pp = ((float4*)p)[i4], mm = ((float4*)m)[i4], vv = ((float4*)v)[i4];
    const float4 gg = ((const float4*)g)[i4];
#define SRL_ADAM1(c)                                                   \
    mm.c = mm.c + (gg.c - mm.c) * (1.0f - b1);                         \
    vv.c = vv.c + (gg.c * gg.c - vv.c) * (1.0f - b2);                  \
    pp.c = pp.c - (lrt * mm.c) / (sqrtf(vv.c) + eps);
    SRL_ADAM1(x) SRL_ADAM1(y) SRL_ADAM1(z) SRL_ADAM1(w)
#undef SRL_ADAM1
    ((float4*)p)[i4] = pp; ((float4*)m)[i4] = mm; ((float4*)v)[i4] = vv;
  } else if (i4 == n4) {
    for (int64_t i = n4 << 2; i < n; ++i) {
      const float gi = g[i];
      const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
      const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
      m[i] = mi; v[i] = vi;
      p[i] = p[i] - (lrt * mi) / (sqrtf(vi) + eps);
    }
  }
}

// ------------------------------------------------------------------------------------------------ K7 Gumbel top-k
// key_i = alpha logit_i - log(-log(u_i)) (logit = -inf stays -inf: an unsampleable slot, memory.py:161); the k largest
// keys in descending order, the lower index first among equal keys.  Stage 1: every workgroup keeps its CHUNK keys in
// registers and extracts its k largest by k block-wide arg-max rounds; stage 2: one workgroup does the same over the
// G k candidates.  alpha is read from device memory (a schedule value that a hipGraph replay must follow).
#define SRL_TOPK_T 256
#define SRL_TOPK_PER 8
#define SRL_TOPK_CHUNK (SRL_TOPK_T * SRL_TOPK_PER)

__global__ void __launch_bounds__(SRL_TOPK_T) k_gumbel_topk1(const float* __restrict__ logits, const float* __restrict__ u,
    const float* __restrict__ alpha_p, int n, int k, float* __restrict__ cand_key, int32_t* __restrict__ cand_idx) {
  __shared__ uint64_t sh[SRL_TOPK_T / 64];
  const int tid = threadIdx.x, base = blockIdx.x * SRL_TOPK_CHUNK;
  const float alpha = alpha_p[0];
  float key[SRL_TOPK_PER];
#pragma unroll
  for (int j = 0; j < SRL_TOPK_PER; ++j) {
    const int i = base + j * SRL_TOPK_T + tid;
    key[j] = -INFINITY;
    if (i < n) {
      const float l = logits[i];
      if (!isinf(l)) key[j] = alpha * l - logf(-logf(u[i]));
    }
  }
  for (int r = 0; r < k; ++r) {
    uint64_t best = 0;
#pragma unroll
    for (int j = 0; j < SRL_TOPK_PER; ++j) {
      const uint64_t kk = vkey(key[j], (uint32_t)(base + j * SRL_TOPK_T + tid));
      best = kk > best ? kk : best;
    }
    best = block_max_u64<SRL_TOPK_T>(best, sh);
    const uint32_t idx = ~(uint32_t)best;
#pragma unroll
    for (int j = 0; j < SRL_TOPK_PER; ++j)
      if ((uint32_t)(base + j * SRL_TOPK_T + tid) == idx) {   // exactly one (thread, j) owns the winner
        // a round that finds only -inf keys (fewer than k sampleable slots in the chunk) reports no candidate
        const bool none = key[j] == -INFINITY;
        cand_key[blockIdx.x * k + r] = key[j];
        cand_idx[blockIdx.x * k + r] = none ? -1 : (int32_t)idx;
        key[j] = -INFINITY;
      }
  }
}

__global__ void __launch_bounds__(SRL_TOPK_T) k_gumbel_topk2(const float* __restrict__ cand_key, const int32_t* __restrict__ cand_idx,
    int nc, int k, int64_t* __restrict__ out_idx, float* __restrict__ out_key) {
  __shared__ uint64_t sh[SRL_TOPK_T / 64];
  const int tid = threadIdx.x;
  for (int r = 0; r < k; ++r) {
    // candidates taken in earlier rounds are recognised by position: pos < 0 marks them
    uint64_t best = 0; int bpos = -1;
    for (int c = tid; c < nc; c += SRL_TOPK_T) {
      const int32_t id = cand_idx[c];
      if (id < 0) continue;
      const uint64_t kk = vkey(cand_key[c], (uint32_t)id);
      if (kk > best) { best = kk; bpos = c; }
    }
    const uint64_t win = block_max_u64<SRL_TOPK_T>(best, sh);
    if (win == 0) {                      // fewer than k sampleable transitions: key -inf flags the entry (memory.py:227-230)
      if (tid == 0) { out_idx[r] = 0; out_key[r] = -INFINITY; }
    } else if (best == win && bpos >= 0) {      // unique: keys carry the (distinct) index
      out_idx[r] = (int64_t)(uint32_t)~(uint32_t)win;
      out_key[r] = cand_key[bpos];
      ((int32_t*)cand_idx)[bpos] = -1;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ K8 scatter / gather
// rows of `bytes` bytes (a multiple of 16) are copied 16 bytes per lane; blockIdx.y = transition
__global__ void __launch_bounds__(256) k_replay_scatter(const uint8_t* __restrict__ s0, const uint8_t* __restrict__ s1,
    int64_t bytes0, int64_t bytes1, const float* __restrict__ reward, const uint8_t* __restrict__ terminal,
    const int64_t* __restrict__ action, int B, int64_t slot, int64_t part_len, uint8_t* __restrict__ m0,
    uint8_t* __restrict__ m1, float* __restrict__ m_reward, uint8_t* __restrict__ m_terminal,
    int64_t* __restrict__ m_action, float* __restrict__ m_logits) {
  const int b = blockIdx.y;
  const int64_t row = (int64_t)b * part_len + slot;                        // memory.py:153
  const int64_t v0 = bytes0 >> 4, v1 = bytes1 >> 4;
  const uint4* a0 = (const uint4*)(s0 + (size_t)b * bytes0);
  const uint4* a1 = (const uint4*)(s1 + (size_t)b * bytes1);
  uint4* d0 = (uint4*)(m0 + (size_t)row * bytes0);
  uint4* d1 = (uint4*)(m1 + (size_t)row * bytes1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < v0 + v1; i += (int64_t)gridDim.x * 256) {
    if (i < v0) d0[i] = a0[i]; else d1[i - v0] = a1[i - v0];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    m_reward[row] = reward[b]; m_terminal[row] = terminal[b]; m_action[row] = action[b];
    m_logits[row] = -INFINITY;                                             // unsampleable until its successor exists (memory.py:161)
  }
}

// blockIdx.y = sample j: rows idx[j] and next[j] of both state tensors, the scalars, and the importance weight
// exp(beta alpha (min_logit - logit)) (memory.py:257-260); alpha, beta, min_logit are device scalars
__global__ void __launch_bounds__(256) k_replay_gather(const int64_t* __restrict__ idx, int64_t part_len, int64_t n_steps,
    int literal_next, int64_t* __restrict__ o_next, const uint8_t* __restrict__ m0, const uint8_t* __restrict__ m1, int64_t bytes0, int64_t bytes1,
    const float* __restrict__ m_reward, const uint8_t* __restrict__ m_terminal, const int64_t* __restrict__ m_action,
    const float* __restrict__ m_logits, const float* __restrict__ alpha_p, const float* __restrict__ beta_p,
    const float* __restrict__ min_logit_p, uint8_t* __restrict__ o_s0, uint8_t* __restrict__ o_s1,
    uint8_t* __restrict__ o_n0, uint8_t* __restrict__ o_n1, int64_t* __restrict__ o_action, float* __restrict__ o_reward,
    uint8_t* __restrict__ o_terminal, float* __restrict__ o_weight) {
  const int j = blockIdx.y;
  const int64_t i = idx[j];
  // the transition n steps on, inside the same partition (memory.py:239-242 as written leaves the partition for every
  // env but the first: `literal_next` reproduces that formula)
  const int64_t n = literal_next ? (i + n_steps) % part_len + i / part_len
                                 : (i % part_len + n_steps) % part_len + (i / part_len) * part_len;
  const int64_t v0 = bytes0 >> 4, v1 = bytes1 >> 4, per = v0 + v1;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < 2 * per; t += (int64_t)gridDim.x * 256) {
    const bool second = t >= per;
    const int64_t tt = second ? t - per : t, row = second ? n : i;
    if (tt < v0) ((uint4*)((second ? o_n0 : o_s0) + (size_t)j * bytes0))[tt] = ((const uint4*)(m0 + (size_t)row * bytes0))[tt];
    else ((uint4*)((second ? o_n1 : o_s1) + (size_t)j * bytes1))[tt - v0] = ((const uint4*)(m1 + (size_t)row * bytes1))[tt - v0];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (o_next) o_next[j] = n;
    o_action[j] = m_action[i];
    o_reward[j] = m_reward[n];
    o_terminal[j] = m_terminal[n];
    if (o_weight) o_weight[j] = expf(beta_p[0] * alpha_p[0] * (min_logit_p[0] - m_logits[i]));
  }
}


// ------------------------------------------------------------------------------------------------ priority trackers
// (max logit, its lowest index) over all rows and (min finite logit, its lowest index; +inf, 0 when there is none) — the two
// scans `ReplayMemory` needs when a tracked slot was overwritten (memory.py:164-177, :282-316).  Two launches: per-block
// partials, then one block over them — the sums of the framework's single-launch multi-block reductions are what returned
// garbage under the concurrent env step; ties go to the lower index.
struct Ext { float mx; long long imx; float mn; long long imn; };
__device__ __forceinline__ void ext_take(Ext& a, const Ext& b) {
  if (b.mx > a.mx || (b.mx == a.mx && b.imx < a.imx)) { a.mx = b.mx; a.imx = b.imx; }
  if (b.mn < a.mn || (b.mn == a.mn && b.imn < a.imn)) { a.mn = b.mn; a.imn = b.imn; }
}
__device__ __forceinline__ Ext ext_block(Ext e, Ext* sh) {
  sh[threadIdx.x] = e;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) { Ext a = sh[threadIdx.x]; ext_take(a, sh[threadIdx.x + s]); sh[threadIdx.x] = a; }
    __syncthreads();
  }
  return sh[0];
}
__global__ void __launch_bounds__(256) k_logit_extrema(const float* __restrict__ logits, long long n, Ext* __restrict__ part) {
  __shared__ Ext sh[256];
  Ext e; e.mx = -INFINITY; e.imx = 0x7fffffffffffffffLL; e.mn = INFINITY; e.imn = 0x7fffffffffffffffLL;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = logits[i];
    Ext o; o.mx = v; o.imx = i; o.mn = (v - v == 0.0f) ? v : INFINITY; o.imn = i;    // finite <=> v - v == 0
    ext_take(e, o);
  }
  const Ext r = ext_block(e, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_logit_extrema_finish(const Ext* __restrict__ part, int nblk, float* __restrict__ out_v,
                                                              long long* __restrict__ out_i) {
  __shared__ Ext sh[256];
  Ext e; e.mx = -INFINITY; e.imx = 0x7fffffffffffffffLL; e.mn = INFINITY; e.imn = 0x7fffffffffffffffLL;
  if ((int)threadIdx.x < nblk) e = part[threadIdx.x];
  const Ext r = ext_block(e, sh);
  if (threadIdx.x == 0) { out_v[0] = r.mx; out_i[0] = r.imx; out_v[1] = r.mn; out_i[1] = r.imn; }
}

}  // namespace

extern "C" {

const char* srl_learner_last_error(void) { return g_err; }

int srl_td_epilogue(const float* q, const float* qn_online, const float* qn_target, const int64_t* actions,
                    const float* rewards, const uint8_t* terminal, const float* weights, float gamma, float huber_delta,
                    float reward_scale, int32_t use_double, float prio_eps, int32_t mb, int32_t A, float* out_loss,
                    float* out_mtd, float* td_abs, float* logits, float* grad_q, float* scratch, int32_t* ticket,
                    void* stream) {
  if (!q || !qn_target || (use_double && !qn_online) || !actions || !rewards || !terminal || !out_loss || !out_mtd ||
      !td_abs || !logits || !scratch || !ticket || mb < 1 || A < 1)
    return fail("srl_td_epilogue: bad argument");
  hipLaunchKernelGGL((k_td_epilogue<256>), dim3(mb), dim3(256), 0, (hipStream_t)stream, q, qn_online, qn_target, actions,
                     rewards, terminal, weights, gamma, huber_delta, reward_scale, (int)use_double, prio_eps, (int)mb, (int)A,
                     out_loss, out_mtd, td_abs, logits, grad_q, scratch, (int*)ticket);
  HIPCHK(hipGetLastError());
  return 0;
}

int srl_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float* state, float lr, float beta1,
                  float beta2, float eps, void* stream) {
  if (!params || !grads || !m || !v || !state || n < 1) return fail("srl_adam_step: bad argument");
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) return fail("srl_adam_step: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr, beta1, beta2);
  const int64_t groups = (n >> 2) + 1;
  hipLaunchKernelGGL(k_adam_step, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, m,
                     v, n, (const float*)state, beta1, beta2, eps);
  HIPCHK(hipGetLastError());
  return 0;
}

int64_t srl_gumbel_topk_scratch_bytes(int64_t n, int32_t k) {
  if (n < 1 || k < 1) return -1;
  const int64_t G = (n + SRL_TOPK_CHUNK - 1) / SRL_TOPK_CHUNK;
  return G * k * (int64_t)(sizeof(float) + sizeof(int32_t));
}

int srl_gumbel_topk(const float* logits, const float* u, const float* alpha, int64_t n, int32_t k, int64_t* out_idx,
                    float* out_key, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!logits || !u || !alpha || !out_idx || !out_key || !scratch || n < 1 || k < 1 || k > SRL_TOPK_CHUNK || n > (1ll << 31) - 1)
    return fail("srl_gumbel_topk: bad argument");
  if (scratch_bytes < srl_gumbel_topk_scratch_bytes(n, k)) return fail("srl_gumbel_topk: scratch too small");
  const int G = (int)((n + SRL_TOPK_CHUNK - 1) / SRL_TOPK_CHUNK);
  float* ck = (float*)scratch;
  int32_t* ci = (int32_t*)(ck + (size_t)G * k);
  hipLaunchKernelGGL(k_gumbel_topk1, dim3(G), dim3(SRL_TOPK_T), 0, (hipStream_t)stream, logits, u, alpha, (int)n, (int)k, ck, ci);
  hipLaunchKernelGGL(k_gumbel_topk2, dim3(1), dim3(SRL_TOPK_T), 0, (hipStream_t)stream, (const float*)ck, (const int32_t*)ci,
                     G * k, (int)k, out_idx, out_key);
  HIPCHK(hipGetLastError());
  return 0;
}

int srl_replay_scatter(const uint8_t* s0, const uint8_t* s1, int64_t bytes0, int64_t bytes1, const float* reward,
                       const uint8_t* terminal, const int64_t* action, int32_t B, int64_t slot, int64_t part_len,
                       uint8_t* m0, uint8_t* m1, float* m_reward, uint8_t* m_terminal, int64_t* m_action, float* m_logits,
                       void* stream) {
  if (!s0 || !s1 || !reward || !terminal || !action || !m0 || !m1 || !m_reward || !m_terminal || !m_action || !m_logits ||
      B < 1 || slot < 0 || slot >= part_len || (bytes0 & 15) || (bytes1 & 15))
    return fail("srl_replay_scatter: bad argument (rows must be multiples of 16 bytes)");
  const int64_t vec = (bytes0 + bytes1) >> 4;
  const unsigned gx = (unsigned)((vec + 255) / 256 > 16 ? 16 : (vec + 255) / 256);
  hipLaunchKernelGGL(k_replay_scatter, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, s0, s1, bytes0, bytes1, reward, terminal,
                     action, (int)B, slot, part_len, m0, m1, m_reward, m_terminal, m_action, m_logits);
  HIPCHK(hipGetLastError());
  return 0;
}

int srl_replay_gather(const int64_t* idx, int32_t mb, int64_t part_len, int64_t n_steps, int32_t literal_next, int64_t* o_next,
                      const uint8_t* m0, const uint8_t* m1, int64_t bytes0,
                      int64_t bytes1, const float* m_reward, const uint8_t* m_terminal, const int64_t* m_action,
                      const float* m_logits, const float* alpha, const float* beta, const float* min_logit, uint8_t* o_s0,
                      uint8_t* o_s1, uint8_t* o_n0, uint8_t* o_n1, int64_t* o_action, float* o_reward, uint8_t* o_terminal,
                      float* o_weight, void* stream) {
  if (!idx || part_len < 1 || n_steps < 1 || !m0 || !m1 || !m_reward || !m_terminal || !m_action || !m_logits || !o_s0 || !o_s1 || !o_n0 || !o_n1 ||
      !o_action || !o_reward || !o_terminal || mb < 1 || (bytes0 & 15) || (bytes1 & 15) || (o_weight && (!alpha || !beta || !min_logit)))
    return fail("srl_replay_gather: bad argument (rows must be multiples of 16 bytes)");
  const int64_t vec = 2 * ((bytes0 + bytes1) >> 4);
  const unsigned gx = (unsigned)((vec + 255) / 256 > 32 ? 32 : (vec + 255) / 256);
  hipLaunchKernelGGL(k_replay_gather, dim3(gx, mb), dim3(256), 0, (hipStream_t)stream, idx, part_len, n_steps, (int)literal_next, o_next, m0, m1, bytes0, bytes1, m_reward,
                     m_terminal, m_action, m_logits, alpha, beta, min_logit, o_s0, o_s1, o_n0, o_n1, o_action, o_reward, o_terminal,
                     o_weight);
  HIPCHK(hipGetLastError());
  return 0;
}

int64_t srl_logit_extrema_scratch_bytes(void) { return 256 * (int64_t)sizeof(Ext); }

int srl_logit_extrema(const float* logits, int64_t n, float* out_v2, int64_t* out_i2, void* scratch, void* stream) {
  if (!logits || n < 1 || !out_v2 || !out_i2 || !scratch) return fail("srl_logit_extrema: bad argument");
  const int nblk = (int)((n + 1023) / 1024 > 256 ? 256 : (n + 1023) / 1024);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_logit_extrema, dim3(nblk), dim3(256), 0, st, logits, (long long)n, (Ext*)scratch);
  hipLaunchKernelGGL(k_logit_extrema_finish, dim3(1), dim3(256), 0, st, (const Ext*)scratch, nblk, out_v2, (long long*)out_i2);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"
