// stackrl_hip.hip — C-ABI of libstackrl_hip.so (include/stackrl_hip.h): host side.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC (see stackrl_amd/build.py).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/stackrl_hip.h"
#include "settle.hip"
#include "render.hip"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* arg = "") {
  snprintf(g_err, sizeof g_err, fmt, arg);
  return code;
}

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      snprintf(g_err, sizeof g_err, "%s: %s", #expr, hipGetErrorString(_e));             \
      return SRL_EHIP;                                                                   \
    }                                                                                    \
  } while (0)

struct EventPair { hipEvent_t a, b; int which; };

}  // namespace

struct srl_env {
  DevParams P;
  MeshHdr* d_mh = nullptr;
  float4* d_mv = nullptr;
  uchar4* d_mt = nullptr;
  float4* d_mp = nullptr;
  DevParams* d_P = nullptr;   // device copy of P for the settle kernel (re-uploaded whenever P changes)
  bool P_dirty = true;
  float* d_objmap = nullptr;
  uint8_t* d_objmap_u8 = nullptr;
  uchar4* d_me = nullptr;
  uint2* d_codec = nullptr;       // overhead depth codec tabulated over the lattice of fl(FAR - z) (DevParams::codec)
  // where srl_reset sends the reward / done of a reset (zeros nobody reads, utils.py:545-552): owned by the handle and
  // allocated in srl_create, so that srl_reset allocates nothing and handles share no buffer
  float* d_reset_r = nullptr;
  uint8_t* d_reset_d = nullptr;
  int step_threads = 256;
  int step_pp = 1;
  int concurrent_envs = 0;   // srl_set_concurrent_envs: envs stepping on the device at the same time, over all handles
  // launch order of srl_k_step (settle.hip, srl_k_order_*): -1 = by batch size (on when the envs outnumber the resident
  // workgroups), 0 = index order, 1 = highest release first (srl_set_launch_order)
  int order_mode = -1;
  unsigned long long* d_order_keys = nullptr;
  int32_t* d_order = nullptr;
  // staged rock records (render.hip, srl_k_stage -> srl_k_render): [n_envs][episode_length][SRL_STAGE_STRIDE] float4;
  // the explicit-pose hook (srl_render_heightmap) has its own, SRL_MAX_BODIES slots per env, allocated at its first use
  float4* d_stage = nullptr;
  float4* d_stage_ext = nullptr;
  // the records are made by the settle kernel's tail for the envs it steps; after a test hook has moved bodies behind its
  // back (srl_set_body_state, srl_step_simulation) the next render is preceded by srl_k_stage over the whole batch
  bool stage_dirty = false;
  size_t step_lds = 0, render_lds = 0, objmap_lds = 0;
  bool profiling = false;
  std::vector<EventPair> pending;
  std::vector<hipEvent_t> pool;
  float acc_ms[4] = {0, 0, 0, 0};   // 0 settle, 1 render, 2 srl_k_stage (hook / stale records only), 3 the two launch-order kernels
  int acc_n[4] = {0, 0, 0, 0};
};

namespace {

// Manifold slots per env: every pair up to 8 rocks (28), 64 up to 16 rocks, 128 above — one contact point per thread in
// every kernel variant (4 lanes per slot).  Pairs whose AABBs overlap at the same time: at most 36 over 512 random
// 16-rock episodes, 85 over 24 32-rock ones (the oracle's statistics); an env that needs more reports
// SRL_ST_PAIR_OVERFLOW and srl_sync_status fails, as it did at the old cap of 192.
int nslots(int L) {
  int np = L * (L - 1) / 2;
  if (np < 1) np = 1;
  const int cap = L <= 16 ? 64 : 128;
  return np < cap ? np : cap;
}

// derived constants: identical expressions to the oracle's derive() so both sides round alike
int derive(DevParams& P) {
  srl_config& c = P.c;
  if (c.n_envs < 1 || c.episode_length < 1 || c.episode_length > SRL_MAX_BODIES)
    return fail(SRL_EINVAL, "n_envs/episode_length out of range");
  if (c.overhead_res < c.object_res || c.object_res < 2 || c.overhead_res > 256 || (c.overhead_res % 8) != 0)
    return fail(SRL_EINVAL, "bad resolutions (overhead_res must be a multiple of 8, <= 256)");
  if (c.metric < 0 || c.metric > SRL_METRIC_EVAL) return fail(SRL_EINVAL, "Invalid value for argument metric");
  P.px = c.object_max_dimension / (float)c.object_res;
  P.inv_px = (float)c.object_res / c.object_max_dimension;
  P.lin_damp = (float)pow(1.0 - (double)c.linear_damping, (double)c.sim_time_step);
  P.ang_damp = (float)pow(1.0 - (double)c.angular_damping, (double)c.sim_time_step);
  // simulator.py:46 int(MAX_STEP_TIME / time_step).  The host mirror passes the value computed from its double (config.py);
  // for a C caller the quotient of the float32 time step is nudged by 1e-6 of itself so that 0.0125f (23999.9996) gives
  // the reference's 24000
  P.max_substeps = c.max_substeps > 0 ? c.max_substeps : (int)(300.0 / (double)c.sim_time_step * (1.0 + 1e-6));
  int H = c.overhead_res, h = c.object_res;
  P.goal_min_h = h; P.goal_min_w = h; P.goal_max_h = H; P.goal_max_w = H;   // rewarder.py:65-80
  P.goal_size = (int)((double)c.goal_size_ratio * H * H);
  if (P.goal_size <= 0) return fail(SRL_EINVAL, "goal_size_ratio must be a scalar in (0,1]");
  if (P.goal_size / P.goal_max_w > P.goal_min_h) P.goal_min_h = P.goal_size / P.goal_max_w;
  if (P.goal_size / P.goal_min_w < P.goal_max_h) P.goal_max_h = P.goal_size / P.goal_min_w;
  P.goal_z = c.max_z - c.object_max_dimension;                                 // observer.py:378-382
  P.scale = c.reward_scale > 0.0f ? c.reward_scale : (float)c.episode_length;  // rewarder.py:97
  P.AW = H - h + 1;                                                            // env.py:207-211
  P.A = P.AW * P.AW;
  if (c.orientation_freedom < 0 || (1 << c.orientation_freedom) > SRL_MAX_ORIENT)
    return fail(SRL_EINVAL, "orientation_freedom must be in 0..4");
  P.n_orient = 1 << c.orientation_freedom;                                     // observer.py:127
  if (c.ordering_freedom != 0 && c.ordering_freedom != 1) return fail(SRL_EINVAL, "ordering_freedom must be 0 or 1");
  P.n_slots = c.ordering_freedom ? c.episode_length * P.n_orient : P.n_orient;  // env.py:472-480
  for (int i = 0; i < P.n_orient; ++i) {   // inverse of getQuaternionFromEuler([0, 0, i 2 pi / n]) (observer.py:129-139)
    const double half = -0.5 * ((double)i * 2.0 * 3.14159265358979323846 / (double)P.n_orient);
    P.orient_q[i][0] = 0.0f; P.orient_q[i][1] = 0.0f;
    P.orient_q[i][2] = i == 0 ? 0.0f : (float)sin(half); P.orient_q[i][3] = i == 0 ? 1.0f : (float)cos(half);
  }
  // observer.py:259-260 / :274-275 constants, rounded to float32 the way numpy rounds python scalars
  double oz = (double)c.object_max_dimension;
  P.elev_num = (float)((double)SRL_FAR * ((double)SRL_FAR - (double)c.max_z));
  P.obj_c1 = (float)((double)SRL_FAR + oz / 2);
  P.obj_c2 = (float)((double)SRL_FAR * (double)SRL_FAR - (oz / 2) * (oz / 2));
  return SRL_OK;
}

// 9 - 16 rocks in batches of 3,072 envs or more run two waves per env with two contact points per thread and without
// the LDS copy of the local vertices: 35 KB per env, so four workgroups share a CU instead of three — these batches are
// throughput-bound (+6 % at 4,096 envs: 47.5 against 50.6 ms per launch; -1 % at 2,048, -6 % at 1,024, where the launch
// lasts as long as its slowest env).  What counts is the number of envs that step on the device at the same time: a
// caller that splits its batch over several handles says so with srl_set_concurrent_envs (`concurrent`, 0 = this handle
// alone).  SRL_STEP_VARIANT=two_wave | four_wave overrides the choice (parity tests).
bool two_wave_variant(const DevParams& P, int concurrent) {
  const int L = P.c.episode_length;
  if (L <= 8 || L > 16) return false;
  const char* v = getenv("SRL_STEP_VARIANT");
  if (v && !strcmp(v, "two_wave")) return true;
  if (v && !strcmp(v, "four_wave")) return false;
  return (concurrent > P.c.n_envs ? concurrent : P.c.n_envs) >= 3072;
}

// threads per env workgroup of the settle kernel (settle.hip "Variants"): 128 up to 8 rocks and for the two-wave variant of
// 9 - 16 rocks, 256 otherwise
int step_threads_of(const DevParams& P, int concurrent) {
  const int NS = nslots(P.c.episode_length);
  if (4 * NS <= 128 && SRL_GMAXP * P.c.episode_length <= 128) return 128;
  if (two_wave_variant(P, concurrent)) return 128;
  return 256;
}

void layout(DevParams& P, int concurrent) {
  int L = P.c.episode_length;
  P.NS = nslots(L);
  P.NP = L * (L - 1) / 2 > 0 ? L * (L - 1) / 2 : 1;
  int o = 0;
  // xyz vectors are stored with a stride of 4 words so that the kernels move them with 16-byte LDS accesses
  // a body's velocities interleaved, 8 words per body: (v.x, w.x, v.y, w.y, v.z, w.z, -, -) — settle.hip Lds::VW
  P.OFF_V = o; P.OFF_W = o + 1; o += 8 * L;
  P.OFF_X = o; o += 4 * L;
  P.OFF_Q = o; o += 4 * L;
  P.OFF_PX = o; o += 4 * L;
  P.OFF_PQ = o; o += 4 * L;
  P.OFF_MESH = o; o += L;
  P.OFF_GM = o; o += SRL_GM_WORDS * L;
  P.OFF_MAN = o; o += SRL_MAN_WORDS * P.NS;
  P.OFF_SOP = o; o += P.NP;
  P.OFF_POS = o; o += P.NS;
  P.OFF_COL = o; o += P.NS;
  P.BLOB = (o + 3) & ~3;
  int s = 0;
  P.S_R = s; s += 9 * L;
  P.S_IW = s; s += 9 * L;
  P.S_AMIN = s; s += 3 * L;
  P.S_AMAX = s; s += 3 * L;
  P.S_BC = s; s += 8 * L;
  // (during the solve the region holds the ground points' row constants instead: SRL_GMAXP x SRL_CG_WORDS words per body)
  P.S_WV = s; s += (3 * P.VS > SRL_GMAXP * SRL_CG_WORDS ? 3 * P.VS : SRL_GMAXP * SRL_CG_WORDS) * L;
  // above 16 rocks the local vertices are read from the (L2-resident) mesh table instead of an LDS copy: 70 instead of
  // 97 KB per env, so that two workgroups share a CU
  if (L > 16 || two_wave_variant(P, concurrent)) P.S_LV = -1; else { P.S_LV = s; s += 3 * P.VS * L; }
  // the tail of the settle kernels stages the env's rocks for the render kernel (stage.h): one StageLds per wave, laid over
  // the scratch words above, which are dead by then — the words below (colouring scratch, misc, pair table) are not
  {
    const int waves = step_threads_of(P, concurrent) / 64;
    const int need = waves * (int)(sizeof(StageLds) / sizeof(float));
    if (s < need) s = need;
  }
  s = (s + 1) & ~1;
  P.S_USED = s; s += 2 * SRL_MAX_BODIES;   // colouring scratch (uint64 per body); BLOB is a multiple of 4 words
  P.S_MISC = s; s += M_WORDS;
  P.S_PAIR = s; s += P.NP;                 // pair id -> (i | j << 16), filled once per launch (settle.hip pair_word)
  P.LDS_WORDS = P.BLOB + s;
}

hipEvent_t get_event(srl_env* env) {
  if (!env->pool.empty()) { hipEvent_t e = env->pool.back(); env->pool.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

// Per-kernel timing (bench.py's roofline leg): the two events are attached to the dispatch itself (hipExtLaunchKernelGGL),
// so they carry the kernel's own begin / end timestamps on its stream — what rocprofv3 reports for the dispatch — and not
// the two marker packets of hipEventRecord calls around it (about 3 us on a 44 us kernel).
EventPair prof_pair(srl_env* env, int which) {
  EventPair p; p.a = nullptr; p.b = nullptr; p.which = which;
  if (env->profiling) { p.a = get_event(env); p.b = get_event(env); env->pending.push_back(p); }
  return p;
}
#define SRL_LAUNCH(env_, which_, kernel_, grid_, block_, lds_, st_, ...)                                           \
  do {                                                                                                             \
    const EventPair ev_ = prof_pair(env_, which_);                                                                 \
    if (ev_.a) hipExtLaunchKernelGGL(kernel_, grid_, block_, lds_, st_, ev_.a, ev_.b, 0, __VA_ARGS__);             \
    else hipLaunchKernelGGL(kernel_, grid_, block_, lds_, st_, __VA_ARGS__);                                       \
  } while (0)

// Ordered launch (settle.hip, srl_k_order_*): worth its two small kernels only when a launch takes several rounds of
// resident workgroups — 2,048 envs or more on this 256-CU part (at most four env workgroups per CU) — and possible while the
// batch's keys fit the sort's LDS (16,384 envs = 128 KB).
constexpr int SRL_ORDER_MIN_ENVS = 2048, SRL_ORDER_MAX_ENVS = 16384;
bool launch_ordered(const srl_env* env) {
  const int n = env->P.c.n_envs;
  if (n > SRL_ORDER_MAX_ENVS || n < 2 || !env->d_order) return false;
  return env->order_mode == 1 || (env->order_mode < 0 && n >= SRL_ORDER_MIN_ENVS);
}

int launch_step_render(srl_env* env, const int64_t* action, void* obs_map, void* obs_obj, float* reward, uint8_t* done,
                       hipStream_t st, int force_reset) {
  if (!env->d_mh) return fail(SRL_ENOMESH, "srl_load_meshes must be called first");
  const DevParams& P = env->P;
  const int n = P.c.n_envs;
  if (env->P_dirty) {   // only after create / load_meshes / seed (all of which leave the device idle), never in steady state:
    // a blocking copy, so that the kernels below read the parameters whatever stream they run on
    HIP_TRY(hipMemcpy(env->d_P, &env->P, sizeof(DevParams), hipMemcpyHostToDevice));
    env->P_dirty = false;
  }
  const DevParams* dP = env->d_P;
  const int32_t* order = nullptr;
  if (force_reset == 0 && launch_ordered(env)) {   // a placement call of a batch that outnumbers the resident workgroups
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    SRL_LAUNCH(env, 3, srl_k_order_keys, dim3(n), dim3(64), 0, st, dP, action, env->d_order_keys);
    SRL_LAUNCH(env, 3, srl_k_order_sort, dim3(1), dim3(1024), (size_t)np2 * 8, st, (const unsigned long long*)env->d_order_keys, n, np2, env->d_order);
    order = env->d_order;
  }
  float4* stage = force_reset < 0 ? nullptr : env->d_stage;   // (sub-steps only: no render follows, the records go stale)
  if (force_reset < 0) env->stage_dirty = true;
  if (env->step_pp == 0) SRL_LAUNCH(env, 0, srl_k_step, dim3(n), dim3(env->step_threads), env->step_lds, st, dP, action, force_reset, order, stage);
  else if (env->step_pp == 2) SRL_LAUNCH(env, 0, srl_k_step_pp2, dim3(n), dim3(env->step_threads), env->step_lds, st, dP, action, force_reset, order, stage);
  else if (env->step_pp == 3) SRL_LAUNCH(env, 0, srl_k_step_t128, dim3(n), dim3(env->step_threads), env->step_lds, st, dP, action, force_reset, order, stage);
  else SRL_LAUNCH(env, 0, srl_k_step_pp1, dim3(n), dim3(env->step_threads), env->step_lds, st, dP, action, force_reset, order, stage);
  if (force_reset < 0) {   // srl_step_simulation: sub-steps only
    HIP_TRY(hipGetLastError());
    return SRL_OK;
  }
  const int L = P.c.episode_length;
#ifdef SRL_NO_STAGE_TAIL     // (A / B builds only: round 4's path, the staging as a kernel of its own in every step)
  env->stage_dirty = true;
#endif
  if (env->stage_dirty) {   // a test hook moved bodies since the records were made: all of them again, by the kernel
    SRL_LAUNCH(env, 2, srl_k_stage, dim3(n, (L + 3) / 4), dim3(256), 0, st, P, env->d_stage, L, (const float*)nullptr,
               (const int32_t*)nullptr, (const int32_t*)nullptr);
    env->stage_dirty = false;
  }
  SRL_LAUNCH(env, 1, srl_k_render, dim3(n), dim3(SRL_RENDER_THREADS), env->render_lds, st, P, (const float4*)env->d_stage, L,
             (uint8_t*)obs_map, (uint8_t*)obs_obj, reward, done, (const int32_t*)nullptr, (float*)nullptr);
  HIP_TRY(hipGetLastError());
  return SRL_OK;
}

}  // namespace

extern "C" {

const char* srl_last_error(void) { return g_err; }

#ifndef SRL_BUILD_INFO
#define SRL_BUILD_INFO "SRL_BUILD_INFO<unknown|>"
#endif
// how this library was built (stackrl_amd/build.py): "SRL_BUILD_INFO<variant|hash of the sources and flags>"
const char* srl_build_info(void) { static const char info[] = SRL_BUILD_INFO; return info; }

int srl_config_default(srl_config* c) {
  if (!c) return fail(SRL_EINVAL, "null config");
  memset(c, 0, sizeof *c);
  c->n_envs = 1;
  c->episode_length = 30;            // DEFAULT_EPISODE_LENGTH, env.py:20
  c->overhead_res = 128; c->object_res = 32;
  c->object_max_dimension = 0.125f; c->max_z = 0.375f;
  c->sim_time_step = 0.01f; c->gravity = 9.8f; c->velocity_threshold = 0.01f;
  c->smooth_placing = 1; c->max_substeps = 0;
  c->metric = SRL_METRIC_IOU; c->goal_size_ratio = 0.25f; c->reward_scale = 1.0f;
  c->reward_pexp = 2; c->reward_oexp = 2;   // Stack-v0 registry: reward_params=2
  c->solver_iterations = 50; c->collision_margin = 0.001f; c->erp = 0.2f;
  c->friction_rock = 0.6f; c->friction_ground = 0.5f;
  c->linear_damping = 0.04f; c->angular_damping = 0.04f; c->warmstart = 0.1f; c->linear_slop = 1e-5f; c->residual_threshold = 1e-7f;
  c->place_at_com = 1;
  c->orientation_freedom = 0;
  c->ordering_freedom = 0;
  return SRL_OK;
}

int srl_create(const srl_config* cfg, srl_env** out) {
  if (!cfg || !out) return fail(SRL_EINVAL, "null argument");
  srl_env* env = new srl_env();
  memset(&env->P, 0, sizeof env->P);
  env->P.c = *cfg;
  int rc = derive(env->P);
  if (rc) { delete env; return rc; }
  env->P.VS = 4;
  layout(env->P, 0);
  DevParams& P = env->P;
  const int n = P.c.n_envs, res = P.c.overhead_res;
  HIP_TRY(hipMalloc((void**)&P.hdr, sizeof(EnvHdr) * (size_t)n));
  HIP_TRY(hipMalloc((void**)&P.blob, sizeof(float) * (size_t)n * P.BLOB));
  HIP_TRY(hipMalloc((void**)&P.H, sizeof(float) * (size_t)n * res * res));
  HIP_TRY(hipMalloc((void**)&P.flags, sizeof(int32_t)));
  HIP_TRY(hipMalloc((void**)&env->d_P, sizeof(DevParams)));
  HIP_TRY(hipMalloc((void**)&env->d_reset_r, sizeof(float) * 4 * (size_t)n));   // up to 4 rewards per env (metric 'all')
  HIP_TRY(hipMalloc((void**)&env->d_stage, sizeof(float4) * (size_t)n * P.c.episode_length * SRL_STAGE_STRIDE));
  if (n <= SRL_ORDER_MAX_ENVS) {   // launch order of the settle kernel (srl_k_order_*): keys + permutation, owned by the handle
    HIP_TRY(hipMalloc((void**)&env->d_order_keys, sizeof(unsigned long long) * (size_t)n));
    HIP_TRY(hipMalloc((void**)&env->d_order, sizeof(int32_t) * (size_t)n));
    if (n > 8192)
      HIP_TRY(hipFuncSetAttribute((const void*)srl_k_order_sort, hipFuncAttributeMaxDynamicSharedMemorySize, SRL_ORDER_MAX_ENVS * 8));
  }
  HIP_TRY(hipMalloc((void**)&env->d_reset_d, (size_t)n));
  HIP_TRY(hipMemset(P.blob, 0, sizeof(float) * (size_t)n * P.BLOB));
  HIP_TRY(hipMemset(P.H, 0, sizeof(float) * (size_t)n * res * res));
  HIP_TRY(hipMemset(P.flags, 0, sizeof(int32_t)));
  {
    std::vector<EnvHdr> h((size_t)n);
    memset(h.data(), 0, sizeof(EnvHdr) * (size_t)n);
    for (int i = 0; i < n; ++i) { h[i].done = 1; h[i].pending = -1; h[i].ncolour = -1; }   // env.py:219-220
    HIP_TRY(hipMemcpy(P.hdr, h.data(), sizeof(EnvHdr) * (size_t)n, hipMemcpyHostToDevice));
  }
  {   // codec table: one entry per float32 between FAR - max_z and FAR (6,145 at max_z = 0.375) + the empty-pixel entry
    const float nearp = SRL_FAR - P.c.max_z;
    const double span = ((double)SRL_FAR - (double)nearp) * 16384.0;
    if (!(nearp >= 512.0f && span <= 65536.0)) {
      srl_destroy(env);
      return fail(SRL_EINVAL, "max_z must be at most 4 (the depth codec is tabulated over the float32 lattice of 1000 - z)");
    }
    const int nt = (int)span + 1;
    HIP_TRY(hipMalloc((void**)&env->d_codec, sizeof(uint2) * (size_t)(nt + 3)));
    hipLaunchKernelGGL(srl_k_codec_table, dim3((nt + 3 + 255) / 256), dim3(256), 0, 0, P, env->d_codec, nt);
    HIP_TRY(hipDeviceSynchronize());
    P.codec = env->d_codec; P.codec_n = nt;
    uint2 row0, rowg, rowo;   // constants evaluated by the device (DevParams::h_empty ...)
    HIP_TRY(hipMemcpy(&row0, env->d_codec, sizeof(uint2), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&rowg, env->d_codec + nt + 1, sizeof(uint2), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&rowo, env->d_codec + nt + 2, sizeof(uint2), hipMemcpyDeviceToHost));
    memcpy(&P.h_empty, &row0.x, sizeof(float));
    P.b_empty = row0.y; P.gbyte = rowg.x; P.zbyte = rowg.y; P.obj_empty_byte = rowo.x;
    P.walk_di = (4 * SRL_RENDER_THREADS) / res; P.walk_dj = 4 * SRL_RENDER_THREADS - P.walk_di * res;
    P.res_magic = (uint32_t)(0x100000000ull / (unsigned long long)res) + 1u;
  }
  *out = env;
  return SRL_OK;
}

void srl_destroy(srl_env* env) {
  if (!env) return;
  (void)hipDeviceSynchronize();
  for (auto& p : env->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto& e : env->pool) (void)hipEventDestroy(e);
  (void)hipFree(env->P.hdr); (void)hipFree(env->P.blob); (void)hipFree(env->P.H); (void)hipFree(env->P.flags); (void)hipFree(env->d_P);
  (void)hipFree(env->d_mh); (void)hipFree(env->d_mv); (void)hipFree(env->d_mt); (void)hipFree(env->d_mp); (void)hipFree(env->d_objmap); (void)hipFree(env->d_objmap_u8);
  (void)hipFree(env->d_codec); (void)hipFree(env->d_me);
  (void)hipFree(env->d_reset_r); (void)hipFree(env->d_reset_d);
  (void)hipFree(env->d_order_keys); (void)hipFree(env->d_order);
  (void)hipFree(env->d_stage); (void)hipFree(env->d_stage_ext);
  delete env;
}

int srl_load_meshes(srl_env* env, const float* verts, const int32_t* vert_off, const int32_t* tris,
                    const int32_t* tri_off, const float* mass_com, int32_t n_mesh) {
  if (!env || !verts || !vert_off || !tris || !tri_off || !mass_com) return fail(SRL_EINVAL, "null argument");
  if (n_mesh < 1) return fail(SRL_EINVAL, "List of object descriptor files is empty.");   // env.py:103
  DevParams& P = env->P;
  std::vector<MeshHdr> mh((size_t)n_mesh);
  std::vector<float4> mv((size_t)vert_off[n_mesh]);
  std::vector<uchar4> mt((size_t)tri_off[n_mesh]);
  std::vector<float4> mp((size_t)tri_off[n_mesh]);
  std::vector<uchar4> me;
  int vs = 4;
  for (int m = 0; m < n_mesh; ++m) {
    MeshHdr& M = mh[m];
    M.vo = vert_off[m]; M.nv = vert_off[m + 1] - vert_off[m];
    M.to = tri_off[m]; M.nt = tri_off[m + 1] - tri_off[m];
    if (M.nv < 4 || M.nv > SRL_MAX_VERTS || M.nt < 4 || M.nt > SRL_MAX_TRIS)
      return fail(SRL_EINVAL, "mesh exceeds SRL_MAX_VERTS/SRL_MAX_TRIS");
    if (M.nv > vs) vs = M.nv;
    float mass = mass_com[4 * m];
    M.cx = mass_com[4 * m + 1]; M.cy = mass_com[4 * m + 2]; M.cz = mass_com[4 * m + 3];
    float lox = 1e30f, loy = 1e30f, loz = 1e30f, hix = -1e30f, hiy = -1e30f, hiz = -1e30f, r2 = 0.0f;
    for (int k = 0; k < M.nv; ++k) {
      const float* p = verts + 3 * (size_t)(M.vo + k);
      float ax = p[0] - M.cx, ay = p[1] - M.cy, az = p[2] - M.cz;   // COM frame
      mv[(size_t)M.vo + k] = make_float4(ax, ay, az, 0.0f);
      lox = fminf(lox, ax); loy = fminf(loy, ay); loz = fminf(loz, az);
      hix = fmaxf(hix, ax); hiy = fmaxf(hiy, ay); hiz = fmaxf(hiz, az);
      float d2 = fmaf(ax, ax, fmaf(ay, ay, az * az));   // dot(a, a) as the kernels define it
      if (d2 > r2) r2 = d2;
    }
    M.radius = sqrtf(r2);
    for (int k = 0; k < M.nt; ++k) {
      const int32_t* t = tris + 3 * (size_t)(M.to + k);
      for (int j = 0; j < 3; ++j)
        if (t[j] < 0 || t[j] >= M.nv) return fail(SRL_EINVAL, "triangle index out of range");
      mt[(size_t)M.to + k] = make_uchar4((unsigned char)t[0], (unsigned char)t[1], (unsigned char)t[2], 0);
      // face plane in the COM frame: unit normal and offset (same expression order as the solver's dot/cross)
      const float4 a = mv[(size_t)M.vo + t[0]], b = mv[(size_t)M.vo + t[1]], c = mv[(size_t)M.vo + t[2]];
      float ux = b.x - a.x, uy = b.y - a.y, uz = b.z - a.z, wx = c.x - a.x, wy = c.y - a.y, wz = c.z - a.z;
      float nx = fmaf(uy, wz, -(uz * wy)), ny = fmaf(uz, wx, -(ux * wz)), nz = fmaf(ux, wy, -(uy * wx));   // cross
      float len = sqrtf(fmaf(nx, nx, fmaf(ny, ny, nz * nz)));
      if (!(len > 0.0f)) return fail(SRL_EINVAL, "degenerate triangle");
      float il = 1.0f / len;
      nx = nx * il; ny = ny * il; nz = nz * il;
      mp[(size_t)M.to + k] = make_float4(nx, ny, nz, fmaf(nx, a.x, fmaf(ny, a.y, nz * a.z)));
    }
    {   // edge list of the closed triangulated surface (the renderer's silhouette test): edge (a < b) -> its two faces
      std::vector<int> first((size_t)M.nv * M.nv, -1);
      M.eo = (int32_t)me.size();
      for (int k = 0; k < M.nt; ++k) {
        const int32_t* t = tris + 3 * (size_t)(M.to + k);
        for (int j = 0; j < 3; ++j) {
          int a = t[j], b = t[(j + 1) % 3];
          if (a > b) { int tmp = a; a = b; b = tmp; }
          if (a == b) return fail(SRL_EINVAL, "degenerate triangle");
          int& slot = first[(size_t)a * M.nv + b];
          if (slot == -1) slot = k;
          else if (slot >= 0) {
            me.push_back(make_uchar4((unsigned char)a, (unsigned char)b, (unsigned char)slot, (unsigned char)k));
            slot = -2;
          } else return fail(SRL_EINVAL, "mesh is not a closed two-manifold (an edge has more than two faces)");
        }
      }
      for (size_t q = 0; q < first.size(); ++q)
        if (first[q] >= 0) return fail(SRL_EINVAL, "mesh is not a closed surface (an edge has only one face)");
      M.ne = (int32_t)me.size() - M.eo;
    }
    // Bullet's default for hull shapes when the URDF inertia is not requested (simulator.py:300 passes no
    // flags): inertia of the solid box spanned by the AABB (btCompoundShape::calculateLocalInertia restated)
    float lx = hix - lox, ly = hiy - loy, lz = hiz - loz;
    float k12 = mass / 12.0f;
    float Ix = k12 * (ly * ly + lz * lz), Iy = k12 * (lx * lx + lz * lz), Iz = k12 * (lx * lx + ly * ly);
    M.inv_mass = 1.0f / mass;
    M.iix = 1.0f / Ix; M.iiy = 1.0f / Iy; M.iiz = 1.0f / Iz;
  }
  (void)hipFree(env->d_mh); (void)hipFree(env->d_mv); (void)hipFree(env->d_mt); (void)hipFree(env->d_mp); (void)hipFree(env->d_objmap); (void)hipFree(env->d_objmap_u8);
  (void)hipFree(env->d_me);
  env->d_mh = nullptr; env->d_mv = nullptr; env->d_mt = nullptr; env->d_mp = nullptr; env->d_objmap = nullptr; env->d_objmap_u8 = nullptr; env->d_me = nullptr;
  const int r = P.c.object_res;
  HIP_TRY(hipMalloc((void**)&env->d_mh, sizeof(MeshHdr) * mh.size()));
  HIP_TRY(hipMalloc((void**)&env->d_mv, sizeof(float4) * mv.size()));
  HIP_TRY(hipMalloc((void**)&env->d_mt, sizeof(uchar4) * mt.size()));
  HIP_TRY(hipMalloc((void**)&env->d_mp, sizeof(float4) * mp.size()));
  HIP_TRY(hipMalloc((void**)&env->d_objmap, sizeof(float) * (size_t)n_mesh * env->P.n_orient * r * r));
  HIP_TRY(hipMalloc((void**)&env->d_objmap_u8, (size_t)n_mesh * env->P.n_orient * r * r));
  HIP_TRY(hipMemcpy(env->d_mh, mh.data(), sizeof(MeshHdr) * mh.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(env->d_mv, mv.data(), sizeof(float4) * mv.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(env->d_mt, mt.data(), sizeof(uchar4) * mt.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(env->d_mp, mp.data(), sizeof(float4) * mp.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&env->d_me, sizeof(uchar4) * me.size()));
  HIP_TRY(hipMemcpy(env->d_me, me.data(), sizeof(uchar4) * me.size(), hipMemcpyHostToDevice));
  P.me = env->d_me;
  P.mh = env->d_mh; P.mv = env->d_mv; P.mt = env->d_mt; P.mp = env->d_mp; P.objmap = env->d_objmap; P.objmap_u8 = env->d_objmap_u8;
  P.n_mesh = n_mesh;
  P.VS = vs;
  env->P_dirty = true;
  env->stage_dirty = true;     // (records made from another mesh table are stale)
  int old_blob = P.BLOB;
  layout(P, env->concurrent_envs);
  if (P.BLOB != old_blob) return fail(SRL_EINVAL, "internal: blob layout changed");
  env->step_lds = sizeof(float) * (size_t)P.LDS_WORDS;
  // experiment hook (tools / DESIGN.md section 8): SRL_STEP_LDS_PAD_KB pads the settle kernel's LDS request, i.e. lowers the
  // number of env workgroups a CU holds, leaving LDS to kernels that run beside it
  if (const char* pad = getenv("SRL_STEP_LDS_PAD_KB")) env->step_lds += (size_t)atoi(pad) * 1024;
  if (env->step_lds > 160 * 1024) return fail(SRL_EINVAL, "episode_length x mesh size exceeds the 160 KB LDS budget");
  // threads per env / pair-manifold points per thread (settle.hip "Variants")
  // 128 threads up to 8 rocks, 256 up to 16 (one contact point per thread), 256 with two points per thread above
  // (settle.hip "Variants")
  if (4 * P.NS <= 128 && SRL_GMAXP * P.c.episode_length <= 128) { env->step_threads = 128; env->step_pp = 0; }
  else if (two_wave_variant(P, env->concurrent_envs)) { env->step_threads = 128; env->step_pp = 3; }
  else if (4 * P.NS <= 256) { env->step_threads = 256; env->step_pp = 1; }
  else { env->step_threads = 256; env->step_pp = 2; }
  const int res = P.c.overhead_res;
  env->render_lds = render_lds_bytes(res);
  if (env->render_lds > 160 * 1024) return fail(SRL_EINVAL, "overhead_res exceeds the 160 KB LDS budget of the render tile (at most 176)");
  env->objmap_lds = 0;
  HIP_TRY(hipFuncSetAttribute((const void*)srl_k_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->step_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)srl_k_step_pp1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->step_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)srl_k_step_pp2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->step_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)srl_k_step_t128, hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->step_lds));
  HIP_TRY(hipFuncSetAttribute((const void*)srl_k_render, hipFuncAttributeMaxDynamicSharedMemorySize, (int)env->render_lds));
  // K3: object maps of the whole pool, once
  hipLaunchKernelGGL(srl_k_objmap, dim3(n_mesh, P.n_orient), dim3(256), env->objmap_lds, 0, P, env->d_objmap, env->d_objmap_u8);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  if (n_mesh < P.c.episode_length) { /* sampled with replacement, env.py:104-106 */ }
  return SRL_OK;
}

int srl_set_concurrent_envs(srl_env* env, int32_t n_envs_on_device) {
  if (!env) return fail(SRL_EINVAL, "null env");
  if (n_envs_on_device < 0) return fail(SRL_EINVAL, "n_envs_on_device must be >= 0");
  if (env->d_mh) return fail(SRL_EINVAL, "srl_set_concurrent_envs must precede srl_load_meshes");
  env->concurrent_envs = n_envs_on_device;
  return SRL_OK;
}

int srl_set_launch_order(srl_env* env, int32_t mode) {
  if (!env) return fail(SRL_EINVAL, "null env");
  if (mode < -1 || mode > 1) return fail(SRL_EINVAL, "launch order mode must be -1 (by batch size), 0 (index order) or 1 (highest release first)");
  if (mode == 1 && !env->d_order) return fail(SRL_EINVAL, "ordered launch supports at most 16,384 envs per handle");
  env->order_mode = mode;
  return SRL_OK;
}

int srl_seed(srl_env* env, uint32_t seed) {
  if (!env) return fail(SRL_EINVAL, "null env");
  env->P.seed = seed;
  env->P.sample_counter = 0;
  env->P_dirty = true;
  // episode counters restart: the header array makes a round trip through the host as one contiguous blocking copy each
  // way (a strided 2-D copy of one word per env was used here until round 3)
  HIP_TRY(hipDeviceSynchronize());
  const int n = env->P.c.n_envs;
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), env->P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) h[i].episode = 0u;
  HIP_TRY(hipMemcpy(env->P.hdr, h.data(), sizeof(EnvHdr) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  return SRL_OK;
}

int srl_set_script(srl_env* env, const int32_t* mesh_ids, const int32_t* goal_rect) {
  if (!env || !mesh_ids || !goal_rect) return fail(SRL_EINVAL, "null argument");
  const DevParams& P = env->P;
  const int n = P.c.n_envs, L = P.c.episode_length;
  for (size_t k = 0; k < (size_t)n * L; ++k)
    if (mesh_ids[k] < 0 || mesh_ids[k] >= P.n_mesh) return fail(SRL_EINVAL, "script mesh id out of range");
  HIP_TRY(hipDeviceSynchronize());
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < L; ++k) h[i].script_ids[k] = mesh_ids[(size_t)i * L + k];
    for (int k = 0; k < 4; ++k) h[i].script_goal[k] = goal_rect[(size_t)i * 4 + k];
    h[i].has_script = 1;
  }
  HIP_TRY(hipMemcpy(P.hdr, h.data(), sizeof(EnvHdr) * (size_t)n, hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  return SRL_OK;
}

int srl_reset(srl_env* env, void* obs_map, void* obs_obj, void* stream) {
  if (!env || !obs_map || !obs_obj) return fail(SRL_EINVAL, "null argument");
  // reward/done of a reset are zeros (utils.py:545-552); the kernels still need somewhere to write them: the handle's own
  // buffers (srl_create)
  return launch_step_render(env, nullptr, obs_map, obs_obj, env->d_reset_r, env->d_reset_d, (hipStream_t)stream, 1);
}

int srl_step(srl_env* env, const int64_t* action, void* obs_map, void* obs_obj, float* reward, uint8_t* done,
             void* stream) {
  if (!env || !action || !obs_map || !obs_obj || !reward || !done) return fail(SRL_EINVAL, "null argument");
  return launch_step_render(env, action, obs_map, obs_obj, reward, done, (hipStream_t)stream, 0);
}

int srl_step_simulation(srl_env* env, int32_t n_substeps, void* stream) {
  if (!env || n_substeps < 1 || n_substeps > 100000) return fail(SRL_EINVAL, "n_substeps must be in [1, 100000]");
  return launch_step_render(env, nullptr, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream, -n_substeps);
}

int srl_set_body_state(srl_env* env, const float* poses, const float* vel) {
  if (!env) return fail(SRL_EINVAL, "null env");
  const DevParams& P = env->P;
  const int n = P.c.n_envs;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  std::vector<float> blob((size_t)n * P.BLOB);
  HIP_TRY(hipMemcpy(blob.data(), P.blob, sizeof(float) * blob.size(), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) {
    float* gb = blob.data() + (size_t)i * P.BLOB;
    for (int b = 0; b < h[i].nb; ++b) {
      if (poses) {
        const float* p = poses + ((size_t)i * SRL_MAX_BODIES + b) * 8;
        for (int k = 0; k < 3; ++k) gb[P.OFF_X + 4 * b + k] = p[k];
        for (int k = 0; k < 4; ++k) gb[P.OFF_Q + 4 * b + k] = p[3 + k];
      }
      if (vel) {
        const float* p = vel + ((size_t)i * SRL_MAX_BODIES + b) * 8;
        for (int k = 0; k < 3; ++k) { gb[P.OFF_V + 8 * b + 2 * k] = p[k]; gb[P.OFF_W + 8 * b + 2 * k] = p[4 + k]; }
      }
    }
  }
  HIP_TRY(hipMemcpy(P.blob, blob.data(), sizeof(float) * blob.size(), hipMemcpyHostToDevice));
  env->stage_dirty = true;
  return SRL_OK;
}

int srl_sample(srl_env* env, int64_t* action, void* stream) {
  if (!env || !action) return fail(SRL_EINVAL, "null argument");
  env->P.sample_counter += 1;
  const int n = env->P.c.n_envs;
  hipLaunchKernelGGL(srl_k_sample, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, env->P, action);
  HIP_TRY(hipGetLastError());
  return SRL_OK;
}

int srl_sync_status(srl_env* env, void* stream) {
  if (!env) return fail(SRL_EINVAL, "null env");
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  int32_t f = 0;
  HIP_TRY(hipMemcpy(&f, env->P.flags, sizeof f, hipMemcpyDeviceToHost));
  if (f) { const int32_t zero = 0; HIP_TRY(hipMemcpy(env->P.flags, &zero, sizeof zero, hipMemcpyHostToDevice)); }
  if (f & 1) return fail(SRL_EINVAL_ACTION, "Invalid action.");
  if (f & 2) return fail(SRL_ESIM_DIVERGED, "Maximum number of simulator steps reached. This may be caused by incorrect behaviour due to a large time step value");
  if (f & 4) return fail(SRL_ESIM_DIVERGED, "More close rock pairs than manifold slots in some env (SRL_ST_PAIR_OVERFLOW): the step is not valid");
  return SRL_OK;
}

int srl_get_state(srl_env* env, float* poses, int32_t* n_bodies, int32_t* substeps, int32_t* status) {
  if (!env) return fail(SRL_EINVAL, "null env");
  const DevParams& P = env->P;
  const int n = P.c.n_envs;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  std::vector<float> blob;
  if (poses) {
    blob.resize((size_t)n * P.BLOB);
    HIP_TRY(hipMemcpy(blob.data(), P.blob, sizeof(float) * blob.size(), hipMemcpyDeviceToHost));
    memset(poses, 0, sizeof(float) * (size_t)n * SRL_MAX_BODIES * 8);
  }
  for (int i = 0; i < n; ++i) {
    if (poses) {
      const float* gb = blob.data() + (size_t)i * P.BLOB;
      float* p = poses + (size_t)i * SRL_MAX_BODIES * 8;
      for (int b = 0; b < h[i].nb; ++b) {
        for (int k = 0; k < 3; ++k) p[b * 8 + k] = gb[P.OFF_X + 4 * b + k];
        for (int k = 0; k < 4; ++k) p[b * 8 + 3 + k] = gb[P.OFF_Q + 4 * b + k];
        p[b * 8 + 7] = (float)((const int32_t*)gb)[P.OFF_MESH + b];
      }
    }
    if (n_bodies) n_bodies[i] = h[i].nb;
    if (substeps) { substeps[2 * i] = h[i].substeps[0]; substeps[2 * i + 1] = h[i].substeps[1]; }
    if (status) status[i] = h[i].status;
  }
  return SRL_OK;
}

int srl_get_sweeps(srl_env* env, int32_t* sweeps) {
  if (!env || !sweeps) return fail(SRL_EINVAL, "null argument");
  const int n = env->P.c.n_envs;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), env->P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) sweeps[i] = h[i].sweeps;
  return SRL_OK;
}

int srl_get_velocities(srl_env* env, float* vel) {
  if (!env || !vel) return fail(SRL_EINVAL, "null argument");
  const DevParams& P = env->P;
  const int n = P.c.n_envs;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  std::vector<float> blob((size_t)n * P.BLOB);
  HIP_TRY(hipMemcpy(blob.data(), P.blob, sizeof(float) * blob.size(), hipMemcpyDeviceToHost));
  memset(vel, 0, sizeof(float) * (size_t)n * SRL_MAX_BODIES * 8);
  for (int i = 0; i < n; ++i) {
    const float* gb = blob.data() + (size_t)i * P.BLOB;
    float* p = vel + (size_t)i * SRL_MAX_BODIES * 8;
    for (int b = 0; b < h[i].nb; ++b)
      for (int k = 0; k < 3; ++k) { p[b * 8 + k] = gb[P.OFF_V + 8 * b + 2 * k]; p[b * 8 + 4 + k] = gb[P.OFF_W + 8 * b + 2 * k]; }
  }
  return SRL_OK;
}

int srl_get_contacts(srl_env* env, float* max_penetration, int32_t* n_points) {
  if (!env || !max_penetration || !n_points) return fail(SRL_EINVAL, "null argument");
  const int n = env->P.c.n_envs;
  float* d_mp = nullptr; int32_t* d_np = nullptr;
  HIP_TRY(hipMalloc((void**)&d_mp, sizeof(float) * (size_t)n));
  HIP_TRY(hipMalloc((void**)&d_np, sizeof(int32_t) * (size_t)n));
  hipLaunchKernelGGL(srl_k_contacts, dim3((n + 63) / 64), dim3(64), 0, 0, env->P, d_mp, d_np);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(max_penetration, d_mp, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(n_points, d_np, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
  (void)hipFree(d_mp); (void)hipFree(d_np);
  return SRL_OK;
}

int srl_get_maps(srl_env* env, float* height, float* object_map, int32_t* goal_rect) {
  if (!env) return fail(SRL_EINVAL, "null env");
  const DevParams& P = env->P;
  const int n = P.c.n_envs, res = P.c.overhead_res, r = P.c.object_res;
  HIP_TRY(hipDeviceSynchronize());
  if (height) HIP_TRY(hipMemcpy(height, P.H, sizeof(float) * (size_t)n * res * res, hipMemcpyDeviceToHost));
  if (object_map || goal_rect) {
    std::vector<EnvHdr> h((size_t)n);
    HIP_TRY(hipMemcpy(h.data(), P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
      if (goal_rect) for (int k = 0; k < 4; ++k) goal_rect[4 * i + k] = h[i].goal[k];
      if (object_map) {
        const size_t per = (size_t)P.n_orient * r * r;   // every observable orientation of one rock
        // the pending rock, or with ordering freedom every rock still unplaced, then empty maps (observer.py:310-327)
        const int shown = P.c.ordering_freedom ? P.c.episode_length : 1;
        for (int k = 0; k < shown; ++k) {
          float* o = object_map + ((size_t)i * shown + k) * per;
          const int m = P.c.ordering_freedom ? (k < h[i].list_pos ? h[i].ids[k] : -1) : h[i].pending;
          if (m >= 0) {
            HIP_TRY(hipMemcpy(o, env->d_objmap + (size_t)m * per, sizeof(float) * per, hipMemcpyDeviceToHost));
          } else {
            // empty map = elev_object(1.0), evaluated like the kernel does
            float e0 = P.obj_c1 - P.obj_c2 / (SRL_FAR + P.c.object_max_dimension * (0.5f - 1.0f));
            for (size_t kk = 0; kk < per; ++kk) o[kk] = e0;
          }
        }
      }
    }
  }
  return SRL_OK;
}

int srl_get_object_map(srl_env* env, int32_t mesh_id, float* object_map) {
  if (!env || !object_map) return fail(SRL_EINVAL, "null argument");
  if (!env->d_objmap) return fail(SRL_ENOMESH, "srl_load_meshes must be called first");
  if (mesh_id < 0 || mesh_id >= env->P.n_mesh) return fail(SRL_EINVAL, "mesh id out of range");
  const size_t per = (size_t)env->P.n_orient * env->P.c.object_res * env->P.c.object_res;
  HIP_TRY(hipMemcpy(object_map, env->d_objmap + (size_t)mesh_id * per, sizeof(float) * per, hipMemcpyDeviceToHost));
  return SRL_OK;
}

int srl_get_stage_records(srl_env* env, float* records, int64_t n_floats) {
  if (!env || !records) return fail(SRL_EINVAL, "null argument");
  const int64_t need = (int64_t)env->P.c.n_envs * env->P.c.episode_length * SRL_STAGE_STRIDE * 4;
  if (n_floats != need) return fail(SRL_EINVAL, "records must hold n_envs x episode_length x SRL_STAGE_STRIDE x 4 floats");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(records, env->d_stage, sizeof(float) * (size_t)need, hipMemcpyDeviceToHost));
  return SRL_OK;
}

int32_t srl_stage_record_stride(void) { return SRL_STAGE_STRIDE; }

int srl_render_heightmap(srl_env* env, const float* poses, const int32_t* mesh_ids, const int32_t* n_bodies,
                         float* height, void* stream) {
  if (!env || !poses || !mesh_ids || !n_bodies || !height) return fail(SRL_EINVAL, "null argument");
  if (!env->d_mh) return fail(SRL_ENOMESH, "srl_load_meshes must be called first");
  hipStream_t st = (hipStream_t)stream;
  const int n = env->P.c.n_envs;
  if (!env->d_stage_ext)
    HIP_TRY(hipMalloc((void**)&env->d_stage_ext, sizeof(float4) * (size_t)n * SRL_MAX_BODIES * SRL_STAGE_STRIDE));
  SRL_LAUNCH(env, 2, srl_k_stage, dim3(n, SRL_MAX_BODIES / 4), dim3(256), 0, st, env->P, env->d_stage_ext, SRL_MAX_BODIES, poses,
             mesh_ids, n_bodies);
  SRL_LAUNCH(env, 1, srl_k_render, dim3(n), dim3(SRL_RENDER_THREADS), env->render_lds, st, env->P, (const float4*)env->d_stage_ext,
             SRL_MAX_BODIES, (uint8_t*)nullptr, (uint8_t*)nullptr, (float*)nullptr, (uint8_t*)nullptr, n_bodies, height);
  HIP_TRY(hipGetLastError());
  return SRL_OK;
}

#ifdef SRL_STAMPS
// diagnostic build only: per-env accumulated phase ticks (100 MHz wall clock), host array [n][8]
int srl_debug_stamps(srl_env* env, long long* out) {
  const int n = env->P.c.n_envs;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), env->P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) { for (int k = 0; k < 8; ++k) out[12 * i + k] = h[i].stamps[k]; for (int k = 0; k < 4; ++k) out[12 * i + 8 + k] = h[i].stamps2[k]; }
  return SRL_OK;
}
int srl_debug_hwid(srl_env* env, long long* out) {   // [n][2]
  const int n = env->P.c.n_envs;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), env->P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) { out[2 * i] = h[i].hwid[0]; out[2 * i + 1] = h[i].hwid[1]; }
  return SRL_OK;
}
int srl_debug_diag(srl_env* env, long long* out) {   // [n][6]
  const int n = env->P.c.n_envs;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), env->P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) for (int k = 0; k < 6; ++k) out[6 * i + k] = h[i].diag[k];
  return SRL_OK;
}
int srl_debug_rstamps(srl_env* env, long long* out, int reset) {
  const int n = env->P.c.n_envs;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<EnvHdr> h((size_t)n);
  HIP_TRY(hipMemcpy(h.data(), env->P.hdr, sizeof(EnvHdr) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) for (int k = 0; k < 8; ++k) { out[8 * i + k] = h[i].rstamps[k]; if (reset) h[i].rstamps[k] = 0; }
  if (reset) HIP_TRY(hipMemcpy(env->P.hdr, h.data(), sizeof(EnvHdr) * (size_t)n, hipMemcpyHostToDevice));
  return SRL_OK;
}
#endif

int srl_set_profiling(srl_env* env, int32_t enable) {
  if (!env) return fail(SRL_EINVAL, "null env");
  env->profiling = enable != 0;
  return SRL_OK;
}

int srl_get_kernel_times(srl_env* env, float* ms3, int32_t* launches3) {
  if (!env) return fail(SRL_EINVAL, "null env");
  HIP_TRY(hipDeviceSynchronize());
  for (auto& p : env->pending) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { env->acc_ms[p.which] += ms; env->acc_n[p.which] += 1; }
    env->pool.push_back(p.a); env->pool.push_back(p.b);
  }
  env->pending.clear();
  for (int k = 0; k < 3; ++k) {
    if (ms3) ms3[k] = env->acc_ms[k];
    if (launches3) launches3[k] = env->acc_n[k];
    env->acc_ms[k] = 0.0f; env->acc_n[k] = 0;
  }
  return SRL_OK;
}

// the same for the two kernels of the ordered launch (srl_k_order_keys + srl_k_order_sort: two launches per ordered step),
// which run before the settle kernel and are not part of its time
int srl_get_order_kernel_times(srl_env* env, float* ms, int32_t* launches) {
  if (!env) return fail(SRL_EINVAL, "null env");
  if (!env->pending.empty()) return fail(SRL_EINVAL, "call srl_get_kernel_times first (it collects the pending events)");
  if (ms) *ms = env->acc_ms[3];
  if (launches) *launches = env->acc_n[3];
  env->acc_ms[3] = 0.0f; env->acc_n[3] = 0;
  return SRL_OK;
}

// Test hook: the keys and the permutation of the latest ordered launch (settle.hip srl_k_order_*): keys[i] = (inverted release
// height bits << 32 | i) of env i, order[k] = the env workgroup k served.  Synchronises the device.
int srl_get_launch_order(srl_env* env, unsigned long long* keys, int32_t* order) {
  if (!env || !keys || !order) return fail(SRL_EINVAL, "null argument");
  if (!env->d_order) return fail(SRL_EINVAL, "this handle has no launch-order buffers (more than 16,384 envs)");
  const int n = env->P.c.n_envs;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(keys, env->d_order_keys, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(order, env->d_order, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
  return SRL_OK;
}

}  // extern "C"
