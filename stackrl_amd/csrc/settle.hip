// settle.hip — K1 (place / smooth placing / settle) + K4 (Observer.pose) + the episode machine.
//
// One workgroup per env: 128 threads (2 waves) for L <= 8 bodies, 256 threads with one contact point per thread up to
// 16 (128 threads with two points per thread in batches of >= 3,072 envs), 256 threads with two above
// (srl_k_step / _pp1 / _t128 / _pp2).  The env's whole persistent state ("blob": poses,
// velocities, ground and body-body manifolds with their warm-start impulses, slot tables) is loaded into
// LDS once, every sub-step runs out of LDS, and the blob is written back once — HBM traffic per env step
// is 2 x BLOB words regardless of how many sub-steps the stop criterion takes.
//
// At the batch sizes of the reference workloads (1,024 - 4,096 envs per GPU) there are about as many envs
// as SIMDs, and a launch lasts as long as its slowest env (stop criterion simulator.py:322-335), so the
// kernel is organised for the LATENCY of one env's sub-step, not for throughput per lane:
//   lane = (body, vertex)   world vertices from an LDS copy of the local ones (one pass, no loops)
//   lane = body             damping + gravity, rotation, inertia, integration
//   16 lanes = body         AABB + ground manifold (deepest-vertex extraction by xor-shuffle min)
//   lane = pair             AABB broadphase over all i<j pairs, slot release
//   16 lanes = slot         GJK closest points and the face-normal SAT fallback: the support / face scans are
//                           split over the 16 lanes and combined with xor shuffles; manifold refresh /
//                           insert by the group's first lane
//   lane = contact point    sequential impulses: each lane keeps the constants of its three rows (normal +
//                           two friction) in registers for all sweeps; body velocities live in LDS
// A sweep visits "ground" then the contact-graph colours in order (block barrier between phases); inside a
// phase the points of one manifold are consecutive lanes of one wave and take turns in index order
// (wave-synchronous, no barrier), manifolds of one colour share no body — so the parallel sweep is exactly
// the sequential definition (DESIGN.md "settle solver") and results are bit-identical to the CPU oracle.
//
// Reference call sites restated: simulator.py:190-258 (step), :310-341 (_place/_stop/_drop),
// observer.py:392-421 (pose), env.py:233-247 (action unflatten, episode list), env.py:266-293 (reset).
#include "srl_device.h"
#include "srl_kernels.h"
#include "stage.h"

#ifdef SRL_DIAG_JITTER
// Diagnostic build (tests/diag/diag_conc.py): every block barrier is preceded by a pseudo-random, wave-uniform delay, so that the
// waves of an env arrive at it — and leave the code before it — in an order that changes from barrier to barrier.  A missing
// barrier between a write of one wave and a read of another would then show as a result that differs from the oracle's.
__device__ __forceinline__ void srl_jitter_sync() {
  unsigned t = (unsigned)__builtin_readcyclecounter() * 2654435761u + (threadIdx.x >> 6) * 40503u;
  t = __builtin_amdgcn_readfirstlane(t);
  const int n = (t >> 13) & 15;
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  t = t * 1664525u + 1013904223u;
  const int m = (t >> 13) & 15;
  for (int i = 0; i < m; ++i) __builtin_amdgcn_s_sleep(16);
}
#define __syncthreads() srl_jitter_sync()
#endif

// lanes per manifold slot in the narrow phase: 32 in the 128-thread variant of <= 8 rocks, 16 where there are more slots than groups
// (round 5, same box: 1,024 x 16 +1.6 %, 4,096 x 16 +2 %, the headline shape indifferent; bit-identical either way)
#ifndef SRL_GJK_GROUP
#define SRL_GJK_GROUP(T, PP) ((T) == 128 && (PP) == 1 ? 32 : 16)
#endif

// Bodies (i < j) of pair id p = j (j - 1) / 2 + i, computed when the kernel fills its LDS table.  (Until round 3 this
// was a pair of process-wide __constant__ tables that every srl_create rewrote: device state shared by all handles.)
__device__ __forceinline__ int pair_word(int p) {
  int j = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
  while (j * (j - 1) / 2 > p) --j;
  while ((j + 1) * j / 2 <= p) ++j;
  return (p - j * (j - 1) / 2) | (j << 16);
}

// misc words in LDS
enum { M_MODE = 0, M_U, M_V, M_NEXT, M_ZMAX, M_FLAGS, M_NCOL, M_CNT, M_NB, M_PENDING, M_STATUS, M_DONE, M_ORIENT, M_MOVING, M_RES0, M_RES1, M_SOLO, M_WORDS = 20 };

struct Lds {
  float* sm;
  const DevParams* P;
  // the blob / scratch layout of DevParams, read once per kernel and kept in scalar registers (laundered through an
  // empty asm so that the compiler cannot re-load them from memory wherever they are used: a scalar load next to LDS
  // traffic shares its wait counter, and with one wave per SIMD nothing hides its latency)
  int oX, oQ, oV, oPX, oPQ, oMESH, oGM, oMAN, oSOP, oPOS, oCOL, oR, oIW, oAMIN, oAMAX, oBC, oWV, oLV, oMISC, oPAIR, vs3;
  __device__ __forceinline__ void init(float* sm_, const DevParams* P_) {
    sm = sm_; P = P_;
    oX = P->OFF_X; oQ = P->OFF_Q; oV = P->OFF_V; oPX = P->OFF_PX; oPQ = P->OFF_PQ; oMESH = P->OFF_MESH;
    oGM = P->OFF_GM; oMAN = P->OFF_MAN; oSOP = P->OFF_SOP; oPOS = P->OFF_POS; oCOL = P->OFF_COL;
    const int blob = P->BLOB;
    oR = blob + P->S_R; oIW = blob + P->S_IW; oAMIN = blob + P->S_AMIN; oAMAX = blob + P->S_AMAX; oBC = blob + P->S_BC;
    oWV = blob + P->S_WV; oLV = blob + P->S_LV; oMISC = blob + P->S_MISC; oPAIR = blob + P->S_PAIR; vs3 = 3 * P->VS;
#ifndef SRL_NO_LAUNDER
#define SRL_KEEP(x) asm volatile("" : "+s"(x))
    SRL_KEEP(oX); SRL_KEEP(oQ); SRL_KEEP(oV); SRL_KEEP(oPX); SRL_KEEP(oPQ); SRL_KEEP(oMESH); SRL_KEEP(oGM);
    SRL_KEEP(oMAN); SRL_KEEP(oSOP); SRL_KEEP(oPOS); SRL_KEEP(oCOL); SRL_KEEP(oR); SRL_KEEP(oIW); SRL_KEEP(oAMIN);
    SRL_KEEP(oAMAX); SRL_KEEP(oBC); SRL_KEEP(oWV); SRL_KEEP(oLV); SRL_KEEP(oMISC); SRL_KEEP(oPAIR); SRL_KEEP(vs3);
#undef SRL_KEEP
#endif
  }
  __device__ __forceinline__ float* X(int b) const { return sm + oX + 4 * b; }
  __device__ __forceinline__ float* Q(int b) const { return sm + oQ + 4 * b; }
  // a body's velocities, interleaved: (v.x, w.x, v.y, w.y, v.z, w.z, -, -) — a (linear, angular) component pair is a register
  // pair as it leaves LDS (one 16-byte and one 8-byte access per body), which is how the pair rows consume it (point_turn)
  __device__ __forceinline__ float* VW(int b) const { return sm + oV + 8 * b; }
  __device__ __forceinline__ float* PX(int b) const { return sm + oPX + 4 * b; }
  __device__ __forceinline__ float* PQ(int b) const { return sm + oPQ + 4 * b; }
  __device__ __forceinline__ int* MESH() const { return (int*)(sm + oMESH); }
  __device__ __forceinline__ float* GM(int b) const { return sm + oGM + SRL_GM_WORDS * b; }
  __device__ __forceinline__ float* MAN(int s) const { return sm + oMAN + SRL_MAN_WORDS * s; }
  __device__ __forceinline__ int* SOP() const { return (int*)(sm + oSOP); }
  __device__ __forceinline__ int* POS() const { return (int*)(sm + oPOS); }
  __device__ __forceinline__ int* COL() const { return (int*)(sm + oCOL); }
  __device__ __forceinline__ float* R(int b) const { return sm + oR + 9 * b; }
  __device__ __forceinline__ float* IW(int b) const { return sm + oIW + 9 * b; }
  __device__ __forceinline__ float* AMIN(int b) const { return sm + oAMIN + 3 * b; }
  __device__ __forceinline__ float* AMAX(int b) const { return sm + oAMAX + 3 * b; }
  __device__ __forceinline__ float* BC(int b) const { return sm + oBC + 8 * b; }  // inv_mass, ii xyz, radius, nv, vo, mesh
  __device__ __forceinline__ float* WV(int b) const { return sm + oWV + vs3 * b; }
  __device__ __forceinline__ float* LV(int b) const { return sm + oLV + vs3 * b; }
  __device__ __forceinline__ int* MISC() const { return (int*)(sm + oMISC); }
  // bodies (i < j) of pair id p: an LDS table filled once per launch (pair_word)
  __device__ __forceinline__ int* PAIR() const { return (int*)(sm + oPAIR); }
  __device__ __forceinline__ void pair(int p, int& i, int& j) const { const int w = PAIR()[p]; i = w & 0xffff; j = w >> 16; }
};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 F2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32: one rounding per half

__device__ __forceinline__ void ldvw(const Lds& L, int b, v3& v, v3& w) {
  const float4 a = *(const float4*)L.VW(b);
  const float2 c = *(const float2*)(L.VW(b) + 4);
  v = V(a.x, a.z, c.x); w = V(a.y, a.w, c.y);
}
__device__ __forceinline__ v3 ldv(const Lds& L, int b) { const float* p = L.VW(b); return V(p[0], p[2], p[4]); }
__device__ __forceinline__ void stvw(const Lds& L, int b, v3 v, v3 w) {
  *(float4*)L.VW(b) = make_float4(v.x, w.x, v.y, w.y);
  *(float2*)(L.VW(b) + 4) = make_float2(v.z, w.z);
}

// ------------------------------------------------------------------ episode reset (env.py:266-293)
__device__ void goal_from_rng(const DevParams& P, uint32_t key, uint32_t episode, int32_t* rect) {
  int H = P.c.overhead_res;
  uint32_t bbit = srl_rng(key, episode, SRL_STREAM_GOAL, 0) >> 31;
  uint32_t u0 = srl_rng(key, episode, SRL_STREAM_GOAL, 1) >> 8;
  uint32_t u1 = srl_rng(key, episode, SRL_STREAM_GOAL, 2) >> 8;
  uint32_t u2 = srl_rng(key, episode, SRL_STREAM_GOAL, 3) >> 8;
  uint32_t lo = u0 < u1 ? u0 : u1; lo = lo < u2 ? lo : u2;
  uint32_t hi = u0 > u1 ? u0 : u1; hi = hi > u2 ? hi : u2;
  uint32_t X = bbit ? hi : lo;   // Beta(3,1) = max of 3 uniforms, Beta(1,3) = min of 3 (rewarder.py:227-231)
  int h = P.goal_min_h + (int)(((uint64_t)X * (uint64_t)(P.goal_max_h - P.goal_min_h)) >> 24);
  int w = P.goal_size / h;
  if (w < P.goal_min_w) w = P.goal_min_w;
  if (w > P.goal_max_w) w = P.goal_max_w;
  int umax = H - h, vmax = H - w;
  int ulo = umax / 8, uhi = 7 * umax / 8 + 1;
  int vlo = vmax / 8, vhi = 7 * vmax / 8 + 1;
  rect[0] = ulo + (int)srl_rng_below(srl_rng(key, episode, SRL_STREAM_GOAL, 4), (uint32_t)(uhi - ulo));
  rect[1] = vlo + (int)srl_rng_below(srl_rng(key, episode, SRL_STREAM_GOAL, 5), (uint32_t)(vhi - vlo));
  rect[2] = h; rect[3] = w;
}

__device__ void env_reset(const DevParams& P, EnvHdr* h, int e) {
  int L = P.c.episode_length;
  uint32_t key = P.seed + (uint32_t)P.c.env_index_offset + (uint32_t)e;   // utils.py:433
  h->episode += 1;
  if (h->has_script) {
    for (int k = 0; k < L; ++k) h->ids[k] = h->script_ids[k];
    for (int k = 0; k < 4; ++k) h->goal[k] = h->script_goal[k];
    h->has_script = 0;
  } else {
    uint32_t draw = 0;
    for (int k = 0; k < L; ++k) {
      for (;;) {   // env.py:268-272: without replacement unless the pool is smaller than L
        int id = (int)srl_rng_below(srl_rng(key, h->episode, SRL_STREAM_MESH, draw++), (uint32_t)P.n_mesh);
        int dup = 0;
        if (P.n_mesh >= L) for (int j = 0; j < k; ++j) dup |= (h->ids[j] == id);
        if (!dup) { h->ids[k] = id; break; }
      }
    }
    goal_from_rng(P, key, h->episode, h->goal);
  }
  h->nb = 0;
  h->ncolour = -1;
  h->pending = h->ids[0];
  h->list_pos = P.c.ordering_freedom ? L : 1;   // ordering freedom: the rocks still unplaced are ids[0 .. list_pos)
  for (int k = 0; k < 4; ++k) h->prev_metric[k] = 0.0f;   // rewarder.py:191-194
  h->substeps[0] = 0; h->substeps[1] = 0;
  h->status = 0;
  h->done = 0;
}

// ------------------------------------------------------------------ per-body derived state
// lane = body: damping then gravity (btRigidBody::applyDamping, then the external-force impulse), rotation
// matrix and world inverse inertia
__device__ __forceinline__ void body_frame(const Lds& L, int b) {
  const DevParams& P = *L.P;
  const float dt = P.c.sim_time_step;
  v3 v, w;
  ldvw(L, b, v, w);
  v = v * P.lin_damp;
  w = w * P.ang_damp;
  v.z = v.z - P.c.gravity * dt;
  stvw(L, b, v, w);
  const float* bc = L.BC(b);
  q4 q; q.x = L.Q(b)[0]; q.y = L.Q(b)[1]; q.z = L.Q(b)[2]; q.w = L.Q(b)[3];
  m3 R = quat_to_mat(q);
  m3 I = inv_inertia_world(R, V(bc[1], bc[2], bc[3]));
#pragma unroll
  for (int i = 0; i < 9; ++i) { L.R(b)[i] = R.m[i]; L.IW(b)[i] = I.m[i]; }
}

// 16 lanes = body: AABB of the world vertices + ground manifold (the deepest vertices within the breaking threshold,
// at most SRL_GMAXP = 4 — Bullet's manifold size — in (dist, index) order; warm-start impulses carried over by vertex id).  Each lane owns the
// vertices gl, gl + 16, ...; the group repeatedly extracts the minimum (dist, index) with xor shuffles.
__device__ __forceinline__ void body_bounds_ground(const Lds& L, int b, int gl) {
  const DevParams& P = *L.P;
  const float* bc = L.BC(b);
  const int nv = __float_as_int(bc[5]);
  const float radius = bc[4];
  const float m = P.c.collision_margin, thr = 0.02f * radius;
  const float* W = L.WV(b);
  v3 lo = V(1e30f, 1e30f, 1e30f), hi = V(-1e30f, -1e30f, -1e30f);
  float cd[SRL_MAX_VERTS / 16];   // this lane's candidate distances (+inf: not a candidate / already taken)
#pragma unroll
  for (int j = 0; j < SRL_MAX_VERTS / 16; ++j) {
    const int k = gl + 16 * j;
    cd[j] = 3.0e38f;
    if (k < nv) {
      v3 a = ld3(W + 3 * k);
      lo = V(fminf(lo.x, a.x), fminf(lo.y, a.y), fminf(lo.z, a.z));
      hi = V(fmaxf(hi.x, a.x), fmaxf(hi.y, a.y), fmaxf(hi.z, a.z));
      const float d = a.z - m;
      if (d < thr) cd[j] = d;
    }
  }
#define SRL_BOUNDS_STEP(C)                                                                                   \
  lo = V(fminf(lo.x, dpp_f<C>(lo.x)), fminf(lo.y, dpp_f<C>(lo.y)), fminf(lo.z, dpp_f<C>(lo.z)));                \
  hi = V(fmaxf(hi.x, dpp_f<C>(hi.x)), fmaxf(hi.y, dpp_f<C>(hi.y)), fmaxf(hi.z, dpp_f<C>(hi.z)));
  SRL_BOUNDS_STEP(0xB1) SRL_BOUNDS_STEP(0x4E) SRL_BOUNDS_STEP(0x141) SRL_BOUNDS_STEP(0x140)   // (srl_device.h: dpp_f)
#undef SRL_BOUNDS_STEP
  float sd[SRL_GMAXP]; int sk[SRL_GMAXP];
  int ns = 0;
#pragma unroll
  for (int r = 0; r < SRL_GMAXP; ++r) {
    sd[r] = 0.0f; sk[r] = -1;
    if (ns == r) {   // group-uniform: previous round found a candidate
      float bd = 3.0e38f; int bk = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < SRL_MAX_VERTS / 16; ++j)
        if (cd[j] < bd) { bd = cd[j]; bk = gl + 16 * j; }   // ascending k: lowest index wins ties
      // arg min of (distance, lowest index) over the 16 lanes on a packed key (srl_device.h: dpp_u64); the distance
      // itself travels along (the key holds it with -0 folded into +0)
      uint64_t key = ((uint64_t)f2o(bd + 0.0f) << 32) | (uint32_t)bk;
#define SRL_MIN_STEP(C)                                                                                      \
      { const uint64_t ok = dpp_u64<C>(key); const float od = dpp_f<C>(bd);                                        \
        const bool t = ok < key; key = t ? ok : key; bd = t ? od : bd; }
      SRL_MIN_STEP(0xB1) SRL_MIN_STEP(0x4E) SRL_MIN_STEP(0x141) SRL_MIN_STEP(0x140)
#undef SRL_MIN_STEP
      bk = (int)(uint32_t)key;
      if (bd < 3.0e38f) {
        sd[r] = bd; sk[r] = bk; ns = r + 1;
#pragma unroll
        for (int j = 0; j < SRL_MAX_VERTS / 16; ++j)
          if (gl + 16 * j == bk) cd[j] = 3.0e38f;
      }
    }
  }
  if (gl != 0) return;
  const float ex = P.c.collision_margin + 0.01f * radius;
  st3(L.AMIN(b), V(lo.x - ex, lo.y - ex, lo.z - ex));
  st3(L.AMAX(b), V(hi.x + ex, hi.y + ex, hi.z + ex));
  float* g = L.GM(b);
  const int onp = __float_as_int(g[0]);
  int ovid[SRL_GMAXP]; float oin[SRL_GMAXP], ot1[SRL_GMAXP], ot2[SRL_GMAXP];
#pragma unroll
  for (int j = 0; j < SRL_GMAXP; ++j) {
    ovid[j] = j < onp ? __float_as_int(g[SRL_GM_VID + j]) : -1;
    oin[j] = g[SRL_GM_IN + j]; ot1[j] = g[SRL_GM_T1 + j]; ot2[j] = g[SRL_GM_T2 + j];
  }
#pragma unroll
  for (int i = 0; i < SRL_GMAXP; ++i) {
    if (i < ns) {
      float in = 0.0f, t1 = 0.0f, t2 = 0.0f;
#pragma unroll
      for (int j = 0; j < SRL_GMAXP; ++j)
        if (ovid[j] == sk[i]) { in = oin[j]; t1 = ot1[j]; t2 = ot2[j]; }
      g[SRL_GM_VID + i] = __int_as_float(sk[i]); g[SRL_GM_DIST + i] = sd[i];
      g[SRL_GM_IN + i] = in; g[SRL_GM_T1 + i] = t1; g[SRL_GM_T2 + i] = t2;
    }
  }
  g[0] = __int_as_float(ns);
}

// ------------------------------------------------------------------ persistent manifold (lane = slot)
__device__ __forceinline__ void manifold_refresh(float* mp, v3 xa, const m3& Ra, v3 xb, const m3& Rb, float thr) {
  int np = __float_as_int(mp[0]);
  for (int i = np - 1; i >= 0; --i) {
    float* p = mp + 4 + SRL_MP_WORDS * i;
    v3 n = ld3(p + 6);
    v3 wa = mmul_add(Ra, ld3(p), xa);
    v3 wb = mmul_add(Rb, ld3(p + 3), xb);
    float d = dot(wa - wb, n);
    bool drop = d > thr;
    if (!drop) {
      v3 proj = madd(wa, n, -d);
      v3 t = wb - proj;
      drop = dot(t, t) > thr * thr;
    }
    if (drop) {
      const float* last = mp + 4 + SRL_MP_WORDS * (np - 1);
      for (int k = 0; k < SRL_MP_WORDS; ++k) p[k] = last[k];
      np--;
    } else {
      p[9] = d;
    }
  }
  mp[0] = __int_as_float(np);
}

__device__ __forceinline__ int manifold_sort_replace(const float* mp, v3 nla, float ndist) {
  int deep = -1; float maxpen = ndist;
  for (int i = 0; i < 4; ++i) {
    float d = mp[4 + SRL_MP_WORDS * i + 9];
    if (d < maxpen) { deep = i; maxpen = d; }
  }
  v3 l0 = ld3(mp + 4), l1 = ld3(mp + 4 + SRL_MP_WORDS), l2 = ld3(mp + 4 + 2 * SRL_MP_WORDS), l3 = ld3(mp + 4 + 3 * SRL_MP_WORDS);
  float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (deep != 0) { v3 c = cross(nla - l1, l3 - l2); res[0] = dot(c, c); }
  if (deep != 1) { v3 c = cross(nla - l0, l3 - l2); res[1] = dot(c, c); }
  if (deep != 2) { v3 c = cross(nla - l0, l3 - l1); res[2] = dot(c, c); }
  if (deep != 3) { v3 c = cross(nla - l0, l2 - l1); res[3] = dot(c, c); }
  int best = 0; float br = res[0];
  if (res[1] > br) { br = res[1]; best = 1; }
  if (res[2] > br) { br = res[2]; best = 2; }
  if (res[3] > br) { br = res[3]; best = 3; }
  return best;
}

__device__ __forceinline__ void manifold_add(float* mp, v3 la, v3 lb, v3 n, float dist, float thr) {
  int np = __float_as_int(mp[0]);
  float shortest = thr * thr; int near_i = -1;
  for (int i = 0; i < np; ++i) {
    v3 d = ld3(mp + 4 + SRL_MP_WORDS * i) - la;
    float d2 = dot(d, d);
    if (d2 < shortest) { shortest = d2; near_i = i; }
  }
  float* p;
  bool keep_impulses = false;
  if (near_i >= 0) { p = mp + 4 + SRL_MP_WORDS * near_i; keep_impulses = true; }
  else if (np < 4) { p = mp + 4 + SRL_MP_WORDS * np; mp[0] = __int_as_float(np + 1); }
  else p = mp + 4 + SRL_MP_WORDS * manifold_sort_replace(mp, la, dist);
  st3(p, la); st3(p + 3, lb); st3(p + 6, n); p[9] = dist;
  if (!keep_impulses) { p[10] = 0.0f; p[11] = 0.0f; p[12] = 0.0f; }
}

// least-penetration face axis for overlapping hulls; the faces are split over the G lanes of the slot's group
// (lane gl takes faces gl, gl + G, ... of body A then of body B) and the best axis is combined with xor shuffles:
// larger separation wins, ties go to the earlier face in (A faces, B faces) order, as in the sequential scan
template <int G>
__device__ __forceinline__ void sat_faces(const DevParams& P, int mesh_a, const float* VA, int na, int mesh_b,
                                          const float* VB, int nb, v3& pa, v3& pb, v3& nrm, float& dist, int gl) {
  float best = -1e30f; int border = 0x7fffffff, bvert = 0; v3 bn = V(0.0f, 0.0f, 1.0f);
  for (int pass = 0; pass < 2; ++pass) {
    const MeshHdr mh = P.mh[pass ? mesh_b : mesh_a];
    const float* VF = pass ? VB : VA;
    const float* VO = pass ? VA : VB;
    int no = pass ? na : nb;
    for (int t = gl; t < mh.nt; t += G) {
      uchar4 tr = P.mt[mh.to + t];
      v3 a = ld3(VF + 3 * tr.x), b = ld3(VF + 3 * tr.y), c = ld3(VF + 3 * tr.z);
      v3 n = cross(b - a, c - a);
      float l2 = dot(n, n);
      if (l2 < 1e-20f) continue;
      n = n * (1.0f / sqrtf(l2));
      float smin = 1e30f; int kmin = 0;
      for (int k = 0; k < no; ++k) {
        float sd = dot(n, ld3(VO + 3 * k) - a);
        if (sd < smin) { smin = sd; kmin = k; }
      }
      if (smin > best) { best = smin; border = pass * 1024 + t; bvert = kmin; bn = n; }
    }
  }
  static_assert(G == 1 || G == 16 || G == 32, "group reductions are written for 1, 16 or 32 lanes");
  auto take = [&](float ob, int oo, int ov, float nx, float ny, float nz) {
    if (ob > best || (ob == best && oo < border)) { best = ob; border = oo; bvert = ov; bn = V(nx, ny, nz); }
  };
#define SRL_SAT_STEP(C) take(dpp_f<C>(best), dpp_i<C>(border), dpp_i<C>(bvert), dpp_f<C>(bn.x), dpp_f<C>(bn.y), dpp_f<C>(bn.z));
  if (G >= 16) { SRL_SAT_STEP(0xB1) SRL_SAT_STEP(0x4E) SRL_SAT_STEP(0x141) SRL_SAT_STEP(0x140) }
#undef SRL_SAT_STEP
  if (G >= 32) {
    float b0, b1, x0, x1, y0, y1, z0, z1; int o0, o1, v0, v1;
    rows_f(best, b0, b1); rows_i(border, o0, o1); rows_i(bvert, v0, v1);
    rows_f(bn.x, x0, x1); rows_f(bn.y, y0, y1); rows_f(bn.z, z0, z1);
    best = b0; border = o0; bvert = v0; bn = V(x0, y0, z0);
    take(b1, o1, v1, x1, y1, z1);
  }
  // (value selects, not two branches writing through pa / pb: the compiler merged those into stores at a computed
  //  scratch offset)
  const bool face_of_a = border < 1024 || border == 0x7fffffff;
  const v3 pv = ld3((face_of_a ? VB : VA) + 3 * bvert), po = madd(pv, bn, -best);
  nrm = face_of_a ? neg(bn) : bn;
  pa = face_of_a ? po : pv;
  pb = face_of_a ? pv : po;
  dist = best;
}

// G lanes per slot (SRL_GJK_GROUP): GJK runs in lock step on all of them (support scans split), the first lane maintains
// the manifold
template <int G>
__device__ __forceinline__ void narrowphase_slot(const Lds& L, int sl, int gl) {
  const DevParams& P = *L.P;
  int pid = L.POS()[sl];
  int a, b; L.pair(pid, a, b);
  const float* bca = L.BC(a);
  const float* bcb = L.BC(b);
  int na = __float_as_int(bca[5]), nb = __float_as_int(bcb[5]);
  float* mp = L.MAN(sl);
  float mg = P.c.collision_margin;
  float thr = 0.02f * fminf(bca[4], bcb[4]);
  v3 xa = ld3(L.X(a)), xb = ld3(L.X(b));
  m3 Ra = ldm(L.R(a)), Rb = ldm(L.R(b));
#ifdef SRL_STAMPS
  long long _n0 = wall_clock64();
#endif
  if (gl == 0) manifold_refresh(mp, xa, Ra, xb, Rb, thr);
#ifdef SRL_STAMPS
  long long _n1 = wall_clock64();
#endif
  v3 axis = ld3(mp + 1);
  v3 pa, pb, n; float d;
  int cache[3];
  cache[0] = __float_as_int(mp[56]); cache[1] = __float_as_int(mp[57]); cache[2] = __float_as_int(mp[58]);
  int rc = gjk_distance<G>(L.WV(a), na, L.WV(b), nb, axis, cache, (mg + mg) + thr, pa, pb, n, d, gl);
#ifdef SRL_STAMPS
  if (threadIdx.x == 0) { long long _n2 = wall_clock64(); EnvHdr* hh = &L.P->hdr[blockIdx.x]; hh->stamps2[0] += _n1 - _n0; hh->stamps2[1] += _n2 - _n1; hh->stamps2[3] += 1; }
  long long _n3 = wall_clock64();
#endif
  if (rc == 2) {   // every lane of the group got the same rc: the face scan is shared
    sat_faces<G>(P, __float_as_int(bca[7]), L.WV(a), na, __float_as_int(bcb[7]), L.WV(b), nb, pa, pb, n, d, gl);
    if (gl != 0) return;
    rc = 1;
  } else {
    if (gl != 0) return;
    st3(mp + 1, axis);
  }
  mp[56] = __int_as_float(cache[0]); mp[57] = __int_as_float(cache[1]); mp[58] = __int_as_float(cache[2]);
  if (rc == 1) {
    float dist = d - (mg + mg);
    if (dist < thr) {
      v3 sa = madd(pa, n, -mg);
      v3 sb = madd(pb, n, mg);
      manifold_add(mp, mtmul(Ra, sa - xa), mtmul(Rb, sb - xb), n, dist, thr);
    }
  }
#ifdef SRL_STAMPS
  if (threadIdx.x == 0) { EnvHdr* hh = &L.P->hdr[blockIdx.x]; hh->stamps2[2] += wall_clock64() - _n3; }
#endif
}

// ------------------------------------------------------------------ sequential impulses (lane = contact point)
// One solver row in precomputed form: direction d, ca = ra x d, aa = Ia ca (and cb, ab for body B),
// k = effective mass denominator.  Same expression trees as the sequential definition.
struct Row { v3 d, ca, aa, cb, ab; float rk, k; };   // k = effective mass denominator, rk = 1 / k

template <bool HAS_B>
__device__ __forceinline__ Row make_row(v3 d, v3 ra, v3 rb, float ima, const m3& Ia, float imb, const m3& Ib) {
  Row r;
  r.d = d;
  r.ca = cross(ra, d);
  r.aa = mmul(Ia, r.ca);
  r.cb = V(0.0f, 0.0f, 0.0f); r.ab = V(0.0f, 0.0f, 0.0f);
  float k = ima + dot(cross(r.aa, ra), d);
  if (HAS_B) {
    r.cb = cross(rb, d);
    r.ab = mmul(Ib, r.cb);
    k = k + (imb + dot(cross(r.ab, rb), d));
  }
  r.rk = 1.0f / k;
  r.k = k;
  return r;
}

// The pair rows in PAIR form (round 5).  A body's velocities leave LDS as the register pairs (v.c, w.c) (Lds::VW); a row's
// constants are kept as the matching pairs — ka[c] = (d.c, ca.c), ua[c] = (d.c ima, aa.c) for body A, kb[c] = (d.c, cb.c),
// ub[c] = (d.c (-imb), -ab.c) for body B — so that both dot products of a body are one chain of three packed operations, the
// velocity update of a body three packed FMAs, and no register move stands between a read, the rows and the write-back.  (The
// compiler's own pairing of row_solve — (va.c, vb.c), (wa.c, wb.c) across the two bodies — cost 23 moves and two sign flips per
// turn of 89 vector instructions.)  The operations and their order are row_solve's: fma(-ab.c, dl, wb.c) = fma(ab.c, -dl, wb.c)
// exactly (the product's sign is exact), the halves of a packed FMA round once each: bit-identical.
struct PRow { f2 ka[3], kb[3], ua[3], ub[3]; float rk, k; };
struct Vel2 { f2 a[3], b[3]; };   // [c] = (v.c, w.c) of body A / B

__device__ __forceinline__ PRow pack_row(const Row& r, float ima, float imb) {
  PRow q;
  const v3 da = r.d * ima, db = r.d * (-imb);
  q.ka[0] = F2(r.d.x, r.ca.x); q.ka[1] = F2(r.d.y, r.ca.y); q.ka[2] = F2(r.d.z, r.ca.z);
  q.kb[0] = F2(r.d.x, r.cb.x); q.kb[1] = F2(r.d.y, r.cb.y); q.kb[2] = F2(r.d.z, r.cb.z);
  q.ua[0] = F2(da.x, r.aa.x); q.ua[1] = F2(da.y, r.aa.y); q.ua[2] = F2(da.z, r.aa.z);
  q.ub[0] = F2(db.x, -r.ab.x); q.ub[1] = F2(db.y, -r.ab.y); q.ub[2] = F2(db.z, -r.ab.z);
  q.rk = r.rk; q.k = r.k;
  return q;
}

__device__ __forceinline__ void prow_solve(const PRow& r, Vel2& u, float target, float& acc, float lo, float hi, float& res) {
  const f2 sa = fma2(r.ka[0], u.a[0], fma2(r.ka[1], u.a[1], r.ka[2] * u.a[2]));   // (dot(d, va), dot(ca, wa))
  const f2 sb = fma2(r.kb[0], u.b[0], fma2(r.kb[1], u.b[1], r.kb[2] * u.b[2]));   // (dot(d, vb), dot(cb, wb))
  float ra = sa.x + sa.y;
  asm("" : "+v"(ra));   // (kept from the vectoriser: paired with the sum below the two adds cost three moves and a packed add)
  const float vrel = ra - (sb.x + sb.y);
  // (target - vrel) rk as one fused step and the linear impulses as (d m^-1) dl: the products that do not depend on the
  // velocities leave the dependency chain (10 dependent operations per row instead of 12; the oracle evaluates the same)
  float dl = fmaf(-vrel, r.rk, target * r.rk);
  // the accumulated impulse clamped to [lo, hi] as the median of the three (v_med3_f32; the oracle restates its zero
  // handling): one instruction on the solver's dependency chain instead of two compare / select pairs through VCC,
  // each of which costs a lone wave its wait states — 6.68 -> 6.02 ms per launch at the headline shape
  const float na = __builtin_amdgcn_fmed3f(acc + dl, lo, hi);
  dl = na - acc;
  acc = na;
  // res: running maximum of the rows' residuals |delta impulse x k| (Bullet: deltaImpulse / m_jacDiagABInv) — off the
  // velocities' dependency chain
  res = fmaxf(res, fabsf(dl * r.k));
  const f2 d2 = F2(dl, dl);
#pragma unroll
  for (int c = 0; c < 3; ++c) { u.a[c] = fma2(r.ua[c], d2, u.a[c]); u.b[c] = fma2(r.ub[c], d2, u.b[c]); }
}

__device__ __forceinline__ void prow_apply(const PRow& r, Vel2& u, float imp) {
  const f2 d2 = F2(imp, imp);
#pragma unroll
  for (int c = 0; c < 3; ++c) { u.a[c] = fma2(r.ua[c], d2, u.a[c]); u.b[c] = fma2(r.ub[c], d2, u.b[c]); }
}

__device__ __forceinline__ float contact_target(const DevParams& P, float dist) {
  float inv_dt = 1.0f / P.c.sim_time_step;
  float pen = dist + P.c.linear_slop;   // Bullet: penetration = distance + m_linearSlop
  return pen > 0.0f ? -(pen * inv_dt) : -((pen * P.c.erp) * inv_dt);
}

// A contact point owned by one lane for the duration of a sub-step's solve
struct Point {
  PRow n, t1, t2;
  float target, in, i1, i2, ima, imb, mu;
  int a, b;        // bodies (b < 0: ground)
  int colour;      // -1 = ground phase
  int idx;         // position inside its manifold (turn order)
  bool valid;
};

// Ground rows.  The ground's normal is +z and its tangents (plane_space of +z) are -y and +x: a ground row's direction is
// sgn e_axis, so its relative velocity takes one component of the linear velocity and its impulse changes that component
// only (stated so in the oracle too): 13 instead of 25 operations per row, and 8 instead of 17 registers of row constants.
struct GRow { v3 ca, aa; float rk, k; };
struct GPoint {
  GRow n, t1, t2;
  float target, in, i1, i2, ima, mu;
  int a, idx;
  bool valid;
};

__device__ __forceinline__ GRow make_grow(v3 d, v3 ra, float ima, const m3& Ia) {
  GRow r;
  r.ca = cross(ra, d);
  r.aa = mmul(Ia, r.ca);
  const float k = ima + dot(cross(r.aa, ra), d);
  r.rk = 1.0f / k;
  r.k = k;
  return r;
}

template <int AXIS, int SGN>
__device__ __forceinline__ void grow_solve(const GRow& r, float ima, v3& v, v3& w, float target, float& acc, float lo,
                                           float hi, float& res) {
  float& va = AXIS == 0 ? v.x : AXIS == 1 ? v.y : v.z;
  const float vrel = fmaf(r.ca.x, w.x, fmaf(r.ca.y, w.y, fmaf(r.ca.z, w.z, SGN < 0 ? -va : va)));
  float dl = fmaf(-vrel, r.rk, target * r.rk);
  const float na = __builtin_amdgcn_fmed3f(acc + dl, lo, hi);
  dl = na - acc;
  acc = na;
  res = fmaxf(res, fabsf(dl * r.k));
  va = fmaf(SGN < 0 ? -ima : ima, dl, va);
  w = madd(w, r.aa, dl);
}

template <int AXIS, int SGN>
__device__ __forceinline__ void grow_apply(const GRow& r, float ima, v3& v, v3& w, float imp) {
  float& va = AXIS == 0 ? v.x : AXIS == 1 ? v.y : v.z;
  va = fmaf(SGN < 0 ? -ima : ima, imp, va);
  w = madd(w, r.aa, imp);
}

__device__ __forceinline__ GPoint make_ground_point(const Lds& L, int b, int i) {
  const DevParams& P = *L.P;
  GPoint p;
  p.valid = false; p.a = b; p.idx = i;
  p.in = 0.0f; p.i1 = 0.0f; p.i2 = 0.0f; p.target = 0.0f; p.ima = 0.0f; p.mu = 0.0f;
  const float* g = L.GM(b);
  if (i >= __float_as_int(g[0])) return p;
  p.valid = true;
  p.ima = L.BC(b)[0];
  p.mu = P.c.friction_rock * P.c.friction_ground;
  const m3 Ia = ldm(L.IW(b));
  const int vid = __float_as_int(g[SRL_GM_VID + i]);
  const v3 pw = ld3(L.WV(b) + 3 * vid);
  const v3 ra = V(pw.x, pw.y, pw.z - P.c.collision_margin) - ld3(L.X(b));
  v3 n = V(0.0f, 0.0f, 1.0f), t1, t2;
  plane_space(n, t1, t2);   // (0, -1, 0), (1, -0, -0)
  p.n = make_grow(n, ra, p.ima, Ia);
  p.t1 = make_grow(t1, ra, p.ima, Ia);
  p.t2 = make_grow(t2, ra, p.ima, Ia);
  p.target = contact_target(P, g[SRL_GM_DIST + i]);
  p.in = g[SRL_GM_IN + i]; p.i1 = g[SRL_GM_T1 + i]; p.i2 = g[SRL_GM_T2 + i];
  return p;
}

// one turn of a ground point: the body's velocities, three axis rows, write back
template <bool WARM>
__device__ __forceinline__ void ground_turn(const Lds& L, GPoint& p, float& res) {
  const float ws = L.P->c.warmstart;
  v3 v, w;
  ldvw(L, p.a, v, w);
  if (WARM) {
    p.in = p.in * ws; p.i1 = p.i1 * ws; p.i2 = p.i2 * ws;
    grow_apply<2, 1>(p.n, p.ima, v, w, p.in);
    grow_apply<1, -1>(p.t1, p.ima, v, w, p.i1);
    grow_apply<0, 1>(p.t2, p.ima, v, w, p.i2);
  } else {
    grow_solve<2, 1>(p.n, p.ima, v, w, p.target, p.in, 0.0f, 1e30f, res);
    const float lim = p.mu * p.in;
    grow_solve<1, -1>(p.t1, p.ima, v, w, 0.0f, p.i1, -lim, lim, res);
    grow_solve<0, 1>(p.t2, p.ima, v, w, 0.0f, p.i2, -lim, lim, res);
  }
  stvw(L, p.a, v, w);
}

// ---- the ground phase by BODY lanes (round 5).  The (up to 4) ground points of one body share that body's velocities and
// nothing else, so a sweep's ground phase is, per body, a sequence of 4 x 3 rows on one pair (v, w).  As turns of point lanes
// the pair went through LDS between the points (write, read, wait: ~100 cycles a turn, and a lone wave has nothing to hide
// them under); here lane = body keeps (v, w) and the points' accumulated impulses in registers for the whole phase and reads
// the rows' constants — made once per sub-step by the point lanes, as before — from LDS in the order it consumes them
// (reads that depend on nothing the rows compute).  Same rows, same order per body, same expression trees: bit-identical.
#define SRL_CG_WORDS 28      // per ground point: 7 float4 (layout: cg_store)
struct GBody {
  float acc[SRL_GMAXP][3];   // accumulated impulses (normal, tangent 1, tangent 2) of the body's ground points
  float ima;
  int np, b;                 // points in the body's ground manifold (0: none / not a body lane)
  float* pvw;                // the body's velocities in LDS (Lds::VW)
  const float4* rec;         // its points' row constants
};

__device__ __forceinline__ float4* cg_rec(const Lds& L, int b, int i) {   // aliases the world vertices, dead during the solve
  return (float4*)L.WV(0) + (SRL_CG_WORDS / 4) * (SRL_GMAXP * b + i);
}
__device__ __forceinline__ void cg_store(const Lds& L, const GPoint& p) {
  float4* r = cg_rec(L, p.a, p.idx);
  r[0] = make_float4(p.n.ca.x, p.n.ca.y, p.n.ca.z, p.n.rk);
  r[1] = make_float4(p.n.aa.x, p.n.aa.y, p.n.aa.z, p.target * p.n.rk);
  r[2] = make_float4(p.t1.ca.x, p.t1.ca.y, p.t1.ca.z, p.t1.rk);
  r[3] = make_float4(p.t1.aa.x, p.t1.aa.y, p.t1.aa.z, p.n.k);
  r[4] = make_float4(p.t2.ca.x, p.t2.ca.y, p.t2.ca.z, p.t2.rk);
  r[5] = make_float4(p.t2.aa.x, p.t2.aa.y, p.t2.aa.z, p.t1.k);
  r[6] = make_float4(p.t2.k, p.mu, 0.0f * p.t1.rk, 0.0f * p.t2.rk);   // (a friction row's target is 0: `target * rk` as grow_solve forms it)
}

// grow_solve with the product target * rk handed in (trk): the same operations on the velocities' chain
template <int AXIS, int SGN>
__device__ __forceinline__ void grow_solve_t(v3 ca, v3 aa, float rk, float k, float ima, v3& v, v3& w, float trk, float& acc,
                                             float lo, float hi, float& res) {
  float& va = AXIS == 0 ? v.x : AXIS == 1 ? v.y : v.z;
  const float vrel = fmaf(ca.x, w.x, fmaf(ca.y, w.y, fmaf(ca.z, w.z, SGN < 0 ? -va : va)));
  float dl = fmaf(-vrel, rk, trk);
  const float na = __builtin_amdgcn_fmed3f(acc + dl, lo, hi);
  dl = na - acc;
  acc = na;
  res = fmaxf(res, fabsf(dl * k));
  va = fmaf(SGN < 0 ? -ima : ima, dl, va);
  w = madd(w, aa, dl);
}

struct GRec { f4 q[7]; };   // one ground point's row constants (cg_store)

template <bool WARM>
__device__ __forceinline__ void ground_point(const GRec& R, GBody& gb, int i, float ws, v3& v, v3& w, float& res) {
  const f4 q0 = R.q[0], q1 = R.q[1], q2 = R.q[2], q3 = R.q[3], q4 = R.q[4], q5 = R.q[5], q6 = R.q[6];
  if (WARM) {
    gb.acc[i][0] = gb.acc[i][0] * ws; gb.acc[i][1] = gb.acc[i][1] * ws; gb.acc[i][2] = gb.acc[i][2] * ws;
    GRow g; g.rk = 0.0f; g.k = 0.0f;
    g.ca = V(q0.x, q0.y, q0.z); g.aa = V(q1.x, q1.y, q1.z); grow_apply<2, 1>(g, gb.ima, v, w, gb.acc[i][0]);
    g.ca = V(q2.x, q2.y, q2.z); g.aa = V(q3.x, q3.y, q3.z); grow_apply<1, -1>(g, gb.ima, v, w, gb.acc[i][1]);
    g.ca = V(q4.x, q4.y, q4.z); g.aa = V(q5.x, q5.y, q5.z); grow_apply<0, 1>(g, gb.ima, v, w, gb.acc[i][2]);
  } else {
    grow_solve_t<2, 1>(V(q0.x, q0.y, q0.z), V(q1.x, q1.y, q1.z), q0.w, q3.w, gb.ima, v, w, q1.w, gb.acc[i][0], 0.0f, 1e30f, res);
    const float lim = q6.y * gb.acc[i][0];
    grow_solve_t<1, -1>(V(q2.x, q2.y, q2.z), V(q3.x, q3.y, q3.z), q2.w, q5.w, gb.ima, v, w, q6.z, gb.acc[i][1], -lim, lim, res);
    grow_solve_t<0, 1>(V(q4.x, q4.y, q4.z), V(q5.x, q5.y, q5.z), q4.w, q6.x, gb.ima, v, w, q6.w, gb.acc[i][2], -lim, lim, res);
  }
}

__device__ __forceinline__ GRec cg_load(const float4* rec, int i) {
  GRec R;
#pragma unroll
  for (int k = 0; k < 7; ++k) { const float4 t = rec[7 * i + k]; R.q[k].x = t.x; R.q[k].y = t.y; R.q[k].z = t.z; R.q[k].w = t.w; }
  return R;
}

// ---- the record of point i + 1 requested while point i is computed (AHEAD).  A plain read whose only use sits in the guarded
// block of the next point is sunk into that block by the compiler (at IR level), a volatile one becomes a flat load with a
// wait behind it, an empty asm that pins the value waits for it: so the reads and their waits are written out.  cg_request
// issues the seven 16-byte reads of record I (and passes `tie`, an operand the current point's first operation needs, through,
// so that the reads stand before that point's arithmetic); cg_arrive waits for everything in flight and passes the record and
// `tie` (a result of the current point's last row) through, so that the wait stands behind that arithmetic and the next point's
// in front of nothing it needs.  Every read is waited for in the block that issued it: no register is handed back to the
// allocator with a read still in flight.  (The compiler's own wait counts do not know these reads; they can only wait longer.)
template <int I>
__device__ __forceinline__ void cg_request(unsigned a, GRec& R, float& tie) {
  asm volatile(
      "ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%10\n\tds_read_b128 %2, %8 offset:%11\n\t"
      "ds_read_b128 %3, %8 offset:%12\n\tds_read_b128 %4, %8 offset:%13\n\tds_read_b128 %5, %8 offset:%14\n\t"
      "ds_read_b128 %6, %8 offset:%15"
      : "=&v"(R.q[0]), "=&v"(R.q[1]), "=&v"(R.q[2]), "=&v"(R.q[3]), "=&v"(R.q[4]), "=&v"(R.q[5]), "=&v"(R.q[6]), "+v"(tie)
      : "v"(a), "n"(112 * I), "n"(112 * I + 16), "n"(112 * I + 32), "n"(112 * I + 48), "n"(112 * I + 64), "n"(112 * I + 80),
        "n"(112 * I + 96));
}
// the body's (v, w) and record 0 in one go (the compiler's own read of (v, w) would be waited for with a count that does not
// know the records behind it: everything)
__device__ __forceinline__ void cg_request_first(unsigned avw, unsigned a, f4& a0, f2& a1, GRec& R) {
  asm volatile(
      "ds_read_b128 %0, %9\n\tds_read_b64 %1, %9 offset:16\n\t"
      "ds_read_b128 %2, %10\n\tds_read_b128 %3, %10 offset:16\n\tds_read_b128 %4, %10 offset:32\n\t"
      "ds_read_b128 %5, %10 offset:48\n\tds_read_b128 %6, %10 offset:64\n\tds_read_b128 %7, %10 offset:80\n\t"
      "ds_read_b128 %8, %10 offset:96"
      : "=&v"(a0), "=&v"(a1), "=&v"(R.q[0]), "=&v"(R.q[1]), "=&v"(R.q[2]), "=&v"(R.q[3]), "=&v"(R.q[4]), "=&v"(R.q[5]), "=&v"(R.q[6])
      : "v"(avw), "v"(a));
}
__device__ __forceinline__ void cg_arrive_first(f4& a0, f2& a1, GRec& R) {   // the next record's seven reads stay in flight
  asm volatile("s_waitcnt lgkmcnt(7)"
               : "+v"(a0), "+v"(a1), "+v"(R.q[0]), "+v"(R.q[1]), "+v"(R.q[2]), "+v"(R.q[3]), "+v"(R.q[4]), "+v"(R.q[5]), "+v"(R.q[6]));
}
template <int LEFT>   // LEFT: reads that may stay in flight (the record requested after this one)
__device__ __forceinline__ void cg_arrive(GRec& R, float& tie) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(R.q[0]), "+v"(R.q[1]), "+v"(R.q[2]), "+v"(R.q[3]), "+v"(R.q[4]), "+v"(R.q[5]), "+v"(R.q[6]), "+v"(tie)
               : "n"(LEFT));
}

template <bool WARM, bool AHEAD>
__device__ __forceinline__ void ground_body(const Lds& L, GBody& gb, float& res) {
  static_assert(SRL_CG_WORDS == 28, "cg_request reads seven float4 per record");
  const float ws = L.P->c.warmstart;
  const float4* rec = gb.rec;
  v3 v, w;
  // The guards are nested: the points of a manifold are a prefix.
  static_assert(SRL_GMAXP == 4, "four nested guards below");
  if (!AHEAD) {
    const float4 a0 = *(const float4*)gb.pvw;
    const float2 a1 = *(const float2*)(gb.pvw + 4);
    v = V(a0.x, a0.z, a1.x); w = V(a0.y, a0.w, a1.y);
    ground_point<WARM>(cg_load(rec, 0), gb, 0, ws, v, w, res);        // (a body lane is called with np >= 1)
    if (gb.np > 1) {
      ground_point<WARM>(cg_load(rec, 1), gb, 1, ws, v, w, res);
      if (gb.np > 2) {
        ground_point<WARM>(cg_load(rec, 2), gb, 2, ws, v, w, res);
        if (gb.np > 3) ground_point<WARM>(cg_load(rec, 3), gb, 3, ws, v, w, res);
      }
    }
  } else {
    const unsigned ra = (unsigned)(size_t)rec;   // the LDS byte address (the low word of the generic pointer)
    GRec r0, r1, r2, r3;
    f4 a0; f2 a1;
    float none = 0.0f;
    cg_request_first((unsigned)(size_t)gb.pvw, ra, a0, a1, r0);
    cg_request<1>(ra, r1, none);
    cg_arrive_first(a0, a1, r0);
    v = V(a0.x, a0.z, a1.x); w = V(a0.y, a0.w, a1.y);
    ground_point<WARM>(r0, gb, 0, ws, v, w, res);
    cg_arrive<0>(r1, w.x);
    if (gb.np > 1) {
      cg_request<2>(ra, r2, w.z);
      ground_point<WARM>(r1, gb, 1, ws, v, w, res);
      cg_arrive<0>(r2, w.x);
      if (gb.np > 2) {
        cg_request<3>(ra, r3, w.z);
        ground_point<WARM>(r2, gb, 2, ws, v, w, res);
        cg_arrive<0>(r3, w.x);
        if (gb.np > 3) ground_point<WARM>(r3, gb, 3, ws, v, w, res);
      }
    }
  }
  *(float4*)gb.pvw = make_float4(v.x, w.x, v.y, w.y); *(float2*)(gb.pvw + 4) = make_float2(v.z, w.z);
}

// The same with the body's (v, w) kept in the caller's registers: sub-steps whose wave holds no pair point (half of all sub-steps, 20 -
// 85 % of the slowest envs') run their sweeps as the ground phase alone, and nothing else reads or writes the velocities during the solve.
template <bool WARM, bool AHEAD>
__device__ __forceinline__ void ground_body_keep(const Lds& L, GBody& gb, v3& v, v3& w, float& res) {
  const float ws = L.P->c.warmstart;
  const float4* rec = gb.rec;
  if (!AHEAD) {
    ground_point<WARM>(cg_load(rec, 0), gb, 0, ws, v, w, res);
    if (gb.np > 1) {
      ground_point<WARM>(cg_load(rec, 1), gb, 1, ws, v, w, res);
      if (gb.np > 2) {
        ground_point<WARM>(cg_load(rec, 2), gb, 2, ws, v, w, res);
        if (gb.np > 3) ground_point<WARM>(cg_load(rec, 3), gb, 3, ws, v, w, res);
      }
    }
  } else {
    const unsigned ra = (unsigned)(size_t)rec;
    GRec r0, r1, r2, r3;
    cg_request<0>(ra, r0, w.z);
    cg_request<1>(ra, r1, w.z);
    cg_arrive<7>(r0, w.z);
    ground_point<WARM>(r0, gb, 0, ws, v, w, res);
    cg_arrive<0>(r1, w.x);
    if (gb.np > 1) {
      cg_request<2>(ra, r2, w.z);
      ground_point<WARM>(r1, gb, 1, ws, v, w, res);
      cg_arrive<0>(r2, w.x);
      if (gb.np > 2) {
        cg_request<3>(ra, r3, w.z);
        ground_point<WARM>(r2, gb, 2, ws, v, w, res);
        cg_arrive<0>(r3, w.x);
        if (gb.np > 3) ground_point<WARM>(r3, gb, 3, ws, v, w, res);
      }
    }
  }
}

__device__ __forceinline__ Point make_pair_point(const Lds& L, int sl, int i) {
  const DevParams& P = *L.P;
  Point p;
  p.valid = false; p.a = 0; p.b = 0; p.colour = 0; p.idx = i;
  p.in = 0.0f; p.i1 = 0.0f; p.i2 = 0.0f; p.target = 0.0f; p.ima = 0.0f; p.imb = 0.0f; p.mu = 0.0f;
  if (sl >= P.NS) return p;
  const int pid = L.POS()[sl];
  if (pid < 0) return p;
  const float* mp = L.MAN(sl);
  if (i >= __float_as_int(mp[0])) return p;
  p.valid = true;
  L.pair(pid, p.a, p.b);
  p.colour = L.COL()[sl];
  p.ima = L.BC(p.a)[0]; p.imb = L.BC(p.b)[0];
  p.mu = P.c.friction_rock * P.c.friction_rock;
  const float* q = mp + 4 + SRL_MP_WORDS * i;
  const m3 Ra = ldm(L.R(p.a)), Rb = ldm(L.R(p.b));
  const m3 Ia = ldm(L.IW(p.a)), Ib = ldm(L.IW(p.b));
  const v3 ra = mmul(Ra, ld3(q));
  const v3 rb = mmul(Rb, ld3(q + 3));
  v3 n = ld3(q + 6), t1, t2;
  plane_space(n, t1, t2);
  p.n = pack_row(make_row<true>(n, ra, rb, p.ima, Ia, p.imb, Ib), p.ima, p.imb);
  p.t1 = pack_row(make_row<true>(t1, ra, rb, p.ima, Ia, p.imb, Ib), p.ima, p.imb);
  p.t2 = pack_row(make_row<true>(t2, ra, rb, p.ima, Ia, p.imb, Ib), p.ima, p.imb);
  p.target = contact_target(P, q[9]);
  p.in = q[10]; p.i1 = q[11]; p.i2 = q[12];
  return p;
}

// one turn of a point: read the velocities of its bodies, three rows, write them back
template <bool WARM>
__device__ __forceinline__ void point_turn(const Lds& L, Point& p, float& res) {
  const float ws = L.P->c.warmstart;
  Vel2 u;
  float* const pa = L.VW(p.a);
  float* const pb = L.VW(p.b);
  {
    const float4 a0 = *(const float4*)pa, b0 = *(const float4*)pb;
    const float2 a1 = *(const float2*)(pa + 4), b1 = *(const float2*)(pb + 4);
    u.a[0] = F2(a0.x, a0.y); u.a[1] = F2(a0.z, a0.w); u.a[2] = F2(a1.x, a1.y);
    u.b[0] = F2(b0.x, b0.y); u.b[1] = F2(b0.z, b0.w); u.b[2] = F2(b1.x, b1.y);
  }
  if (WARM) {
    p.in = p.in * ws; p.i1 = p.i1 * ws; p.i2 = p.i2 * ws;
    prow_apply(p.n, u, p.in);
    prow_apply(p.t1, u, p.i1);
    prow_apply(p.t2, u, p.i2);
  } else {
    prow_solve(p.n, u, p.target, p.in, 0.0f, 1e30f, res);
    const float lim = p.mu * p.in;
    prow_solve(p.t1, u, 0.0f, p.i1, -lim, lim, res);
    prow_solve(p.t2, u, 0.0f, p.i2, -lim, lim, res);
  }
  *(float4*)pa = make_float4(u.a[0].x, u.a[0].y, u.a[1].x, u.a[1].y); *(float2*)(pa + 4) = make_float2(u.a[2].x, u.a[2].y);
  *(float4*)pb = make_float4(u.b[0].x, u.b[0].y, u.b[1].x, u.b[1].y); *(float2*)(pb + 4) = make_float2(u.b[2].x, u.b[2].y);
}

// A wave runs only the turns it has points for (a turn no lane takes still costs a lone wave its control flow: 8 + 4 ncol empty
// turns per sweep were a third of the sub-step's instructions in round 1; rounds 2 - 5 counted the turns per phase / per colour,
// now a phase ends at its first empty mask).
// A sweep ends with a block barrier; before it every wave that still holds a row whose squared residual exceeds the
// threshold writes the sweep's number `gsweep` (counted over the whole launch, so a stale word never matches) into the
// LDS word of the sweep's parity; after the barrier every thread reads that word: `true` = some row has not converged
// (Bullet's m_leastSquaresResidual > m_leastSquaresResidualThreshold; the maximum over rows is order-independent, so
// the parallel sweep decides exactly as the sequential definition).  The word of the other parity is the one the
// next sweep writes, so a fast wave cannot disturb a slow wave's read.
// SOLO: every contact point of the env sits in wave 0 (the common case with few rocks: the ground points are the first
// lanes, and manifold slots are handed out lowest first), so the sweep is wave 0's alone: the phases follow each other in
// program order — the LDS serves one wave's accesses in order — without a single block barrier, and the residual is a
// ballot.  The other waves skip the sweeps and wait at the barrier that ends the solve.
template <bool WARM, int PP, bool SOLO, bool GB, bool GA>
__device__ __forceinline__ bool solver_sweep(const Lds& L, GPoint& gp, GBody& gb, Point (&pp)[PP], int ncol, int gslot,
                                             const int (&pslot)[PP], int gsweep, const unsigned long long (&cm)[2]) {
  // gslot / pslot: the turn a lane's point takes (ground: its index; colour phases: 4 * colour + index; -1: none)
  float res = 0.0f;
  // A turn's guard is a SCALAR mask: a point's index in its manifold is its lane & 3 (four consecutive lanes per body / per slot), so
  // turn i of a phase is (the phase's lanes) & 0x1111... << i — `s_and_b64` + `s_and_saveexec_b64` on a uniform value
  // (inverse ballot).  As a vector compare per turn inside a rolled loop the guard and the loop's backward branch cost a lone wave
  // ≈ 75 cycles a turn (microbenchmark: compare -> VCC -> exec 32 cycles, a taken branch ≈ 32): the turns are unrolled, a
  // phase ends at its first empty turn (the points of a manifold are a prefix).
  constexpr unsigned long long T0 = 0x1111111111111111ull;
  if (!GB) {     // the points of a body as turns of point lanes (round 1 - 4; kept for the variant that loses with body lanes)
    const unsigned long long gm = __ballot(gslot >= 0);
#pragma unroll
    for (int i = 0; i < SRL_GMAXP; ++i) {
      const unsigned long long t = gm & (T0 << i);
      if (t == 0) break;
      if (__builtin_amdgcn_inverse_ballot_w64(t)) ground_turn<WARM>(L, gp, res);
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    if (gb.np > 0) ground_body<WARM, !WARM && GA>(L, gb, res);
    __builtin_amdgcn_wave_barrier();
  }
  int c0 = 0;
#ifndef SRL_NO_COLOUR_MASKS
  if (SOLO && PP == 1) {   // the first two colours from masks made once per sub-step, straight-line (no ballot, no loop branch per colour)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c >= ncol) break;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned long long t = cm[c] & (T0 << i);
        if (t == 0) break;
        if (__builtin_amdgcn_inverse_ballot_w64(t)) point_turn<WARM>(L, pp[0], res);
        __builtin_amdgcn_wave_barrier();
      }
    }
    c0 = 2;
  }
#endif
#pragma unroll 1
  for (int c = c0; c < ncol; ++c) {
    if (!SOLO) __syncthreads();
    unsigned long long mc[PP];
#pragma unroll
    for (int r = 0; r < PP; ++r) mc[r] = __ballot((pslot[r] >> 2) == c);   // (pslot = -1: no point)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned long long any = 0;
#pragma unroll
      for (int r = 0; r < PP; ++r) any |= mc[r] & (T0 << i);
      if (any == 0) break;
#pragma unroll
      for (int r = 0; r < PP; ++r)
        if (__builtin_amdgcn_inverse_ballot_w64(mc[r] & (T0 << i))) point_turn<WARM>(L, pp[r], res);
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (SOLO) return WARM ? true : __ballot(res * res > L.P->c.residual_threshold) != 0ull;
  int* word = L.MISC() + M_RES0 + (gsweep & 1);
  if (!WARM) {
    if (__ballot(res * res > L.P->c.residual_threshold) != 0ull && (threadIdx.x & 63) == 0) *word = gsweep;
  }
  __syncthreads();
  return WARM ? true : *word == gsweep;
}

// ------------------------------------------------------------------ one sub-step (block-wide)
#ifdef SRL_STAMPS
#define STAMP(k) do { if (tid == 0) { long long _t = wall_clock64(); L.P->hdr[blockIdx.x].stamps[k] += _t - _t0; _t0 = _t; } } while (0)
#else
#define STAMP(k)
#endif

template <int T, int PP>
__device__ __forceinline__ void substep(const Lds& L, int nb, int tid, int& gsweep) {
  const DevParams& P = *L.P;
  // With two contact points per thread the register file is full: lane-dependent LDS addresses that the compiler
  // hoists out of the sub-step loop end up spilled to scratch and are reloaded from memory in every sub-step.  An
  // opaque copy of the thread index keeps them inside the loop, where recomputing one costs an instruction or two.
  if (PP >= 2) asm volatile("" : "+v"(tid));
#ifdef SRL_STAMPS
  long long _t0 = wall_clock64();
#endif
  int* misc = L.MISC();
  // (1) lane = body: damping, gravity, frame
  if (tid < nb) body_frame(L, tid);
  if (tid == 0) { misc[M_FLAGS] = 0; misc[M_SOLO] = 1; }
  __syncthreads();
  STAMP(0);
  // (2) lane = (body, vertex): world vertices
  for (int it = tid; it < nb * P.VS; it += T) {
    const int b = it / P.VS, k = it - b * P.VS;
    if (k < __float_as_int(L.BC(b)[5])) {
      const m3 R = ldm(L.R(b));
      v3 lv;
      if (P.S_LV >= 0) lv = ld3(L.LV(b) + 3 * k);
      else { const float4 g4 = P.mv[__float_as_int(L.BC(b)[6]) + k]; lv = V(g4.x, g4.y, g4.z); }   // no LDS copy: the L2-resident mesh table
      st3(L.WV(b) + 3 * k, mmul_add(R, lv, ld3(L.X(b))));
    }
  }
  __syncthreads();
  STAMP(1);
  // (3) 16 lanes = body: AABB + ground manifold
  for (int b = tid >> 4; b < nb; b += T >> 4) body_bounds_ground(L, b, tid & 15);
  __syncthreads();
  STAMP(2);
  // (4) broadphase: AABB overlap for every pair; release slots of pairs that separated
  const int npair = nb * (nb - 1) / 2;
  int fl = 0;
  for (int pid = tid; pid < npair; pid += T) {
    int i, j; L.pair(pid, i, j);
    v3 ai = ld3(L.AMIN(i)), bi = ld3(L.AMAX(i)), aj = ld3(L.AMIN(j)), bj = ld3(L.AMAX(j));
    bool ov = ai.x <= bj.x && aj.x <= bi.x && ai.y <= bj.y && aj.y <= bi.y && ai.z <= bj.z && aj.z <= bi.z;
    int sl = L.SOP()[pid];
    if (!ov && sl >= 0) { L.SOP()[pid] = -1; L.POS()[sl] = -1; fl |= 1; }
    if (ov && sl < 0) fl |= 2;
  }
  if (fl) atomicOr(&misc[M_FLAGS], fl);
  __syncthreads();
  const int flags = misc[M_FLAGS], ncol0 = misc[M_NCOL];
  if (flags || ncol0 < 0) {
    __syncthreads();   // every thread has taken its snapshot before thread 0 rewrites the words
    if (tid == 0) {
      if (flags & 2) {   // allocate slots for newly close pairs: ascending pair id, lowest free slot first
        for (int pid = 0; pid < npair; ++pid) {
          if (L.SOP()[pid] >= 0) continue;
          int i, j; L.pair(pid, i, j);
          v3 ai = ld3(L.AMIN(i)), bi = ld3(L.AMAX(i)), aj = ld3(L.AMIN(j)), bj = ld3(L.AMAX(j));
          bool ov = ai.x <= bj.x && aj.x <= bi.x && ai.y <= bj.y && aj.y <= bi.y && ai.z <= bj.z && aj.z <= bi.z;
          if (!ov) continue;
          int sl = -1;
          for (int k = 0; k < P.NS; ++k) if (L.POS()[k] < 0) { sl = k; break; }
          if (sl < 0) { misc[M_STATUS] |= SRL_ST_PAIR_OVERFLOW; atomicOr(P.flags, 4); continue; }
          L.SOP()[pid] = sl; L.POS()[sl] = pid;
          float* mp = L.MAN(sl);
          mp[0] = __int_as_float(0);
          mp[56] = __int_as_float(0);   // no cached simplex
          st3(mp + 1, ld3(L.X(i)) - ld3(L.X(j)));
        }
      }
      // greedy colouring in slot order
      uint64_t* used = (uint64_t*)(L.sm + P.BLOB + P.S_USED);   // LDS scratch, 8-byte aligned
      for (int b = 0; b < SRL_MAX_BODIES; ++b) used[b] = 0;
      int nc = 0;
      for (int sl = 0; sl < P.NS; ++sl) {
        int pid = L.POS()[sl];
        if (pid < 0) continue;
        int i, j; L.pair(pid, i, j);
        uint64_t u = used[i] | used[j];
        int c = __ffsll((long long)~u) - 1;
        L.COL()[sl] = c;
        used[i] |= (uint64_t)1 << c;
        used[j] |= (uint64_t)1 << c;
        if (c + 1 > nc) nc = c + 1;
      }
      misc[M_NCOL] = nc;
    }
    __syncthreads();
  }
  const int ncol = misc[M_NCOL];
  STAMP(3);
  // (5) narrowphase: G lanes per slot
  {
    constexpr int G = SRL_GJK_GROUP(T, PP);
    const int gl = tid & (G - 1);
    for (int sl = tid / G; sl < P.NS; sl += T / G)
      if (L.POS()[sl] >= 0) narrowphase_slot<G>(L, sl, gl);
  }
  __syncthreads();
  STAMP(4);
  // (6) sequential impulses: lane = contact point, row constants in registers for all sweeps
  {
    GPoint gp = make_ground_point(L, tid / SRL_GMAXP, tid % SRL_GMAXP);
    if (tid / SRL_GMAXP >= nb) gp.valid = false;
    Point pp[PP];
#pragma unroll
    for (int r = 0; r < PP; ++r) pp[r] = make_pair_point(L, (tid + r * T) >> 2, tid & 3);
    {   // a wave other than the first that holds a contact point rules the solo sweep out
      bool mine = gp.valid;
#pragma unroll
      for (int r = 0; r < PP; ++r) mine |= pp[r].valid;
      if (tid >= 64 && __ballot(mine) != 0ull && (tid & 63) == 0) misc[M_SOLO] = 0;
    }
    __syncthreads();   // every lane has read the pre-solve velocities' companions (R, Iw, manifolds)
    GBody gb;
    gb.np = 0; gb.b = 0; gb.ima = 0.0f;
    gb.pvw = L.VW(0); gb.rec = cg_rec(L, 0, 0);
#pragma unroll
    for (int i = 0; i < SRL_GMAXP; ++i) { gb.acc[i][0] = 0.0f; gb.acc[i][1] = 0.0f; gb.acc[i][2] = 0.0f; }
    // The ground phase by body lanes (ground_body) in every variant but the four-wave one with one point per thread (9 - 16
    // rocks, small batches): built for three waves per SIMD (168 VGPRs) it spills 185 registers with it against 113 and loses
    // 9 % (1,024 / 2,048 envs x 16 rocks); the others gain 3 - 5 % (profiles/r05_ground_body_ab.txt).  SRL_GROUND_TURNS: A / B.
#ifdef SRL_GROUND_TURNS
    constexpr bool GB = false;
#else
    constexpr bool GB = !(T == 256 && PP == 1);
#endif
    // the next ground point's record requested a point ahead (ground_body<AHEAD>): where the variant has the registers
#ifndef SRL_GA_MASK
#define SRL_GA_MASK 1
#endif
    constexpr bool GA = GB && ((T == 128 && PP == 1 && (SRL_GA_MASK & 1)) || (T == 128 && PP == 2 && (SRL_GA_MASK & 2)) ||
                               (T == 256 && PP == 2 && (SRL_GA_MASK & 4)));
    if (GB) {
    // the rows' constants of every ground point to LDS (over the world vertices, which nothing reads before the next
    // sub-step rewrites them), for the body lanes (tid < nb: wave 0)
    if (gp.valid) cg_store(L, gp);
    if (tid < nb) {
      const float* g = L.GM(tid);
      gb.b = tid; gb.np = __float_as_int(g[0]); gb.ima = L.BC(tid)[0];
      gb.pvw = L.VW(tid); gb.rec = cg_rec(L, tid, 0);
#pragma unroll
      for (int i = 0; i < SRL_GMAXP; ++i)
        if (i < gb.np) { gb.acc[i][0] = g[SRL_GM_IN + i]; gb.acc[i][1] = g[SRL_GM_T1 + i]; gb.acc[i][2] = g[SRL_GM_T2 + i]; }
    }
    __syncthreads();   // the records are in LDS
    }
    STAMP(5);
#ifdef SRL_DIAG_NOSOLO
    const bool solo = false;                           // diagnostic build: every sweep through the block-wide path
#else
    const bool solo = PP == 1 && misc[M_SOLO] != 0;   // (the variants with two points per thread are out of registers as it is)
#endif
    const int gslot = gp.valid ? gp.idx : -1;
    // the lanes of the first two colours (solo sweeps: solver_sweep)
    unsigned long long cm[2];
    cm[0] = __ballot(pp[0].valid && pp[0].colour == 0); cm[1] = __ballot(pp[0].valid && pp[0].colour == 1);
    int pslot[PP];
#pragma unroll
    for (int r = 0; r < PP; ++r) pslot[r] = pp[r].valid ? 4 * pp[r].colour + pp[r].idx : -1;
    // at most solver_iterations sweeps, ended early once a sweep's largest squared residual is <= the threshold
    // (btSequentialImpulseConstraintSolver::solveGroupCacheFriendlyIterations)
    if (solo) {
      int done = 0;
      if (tid < 64) {
        bool anyp = false;
#pragma unroll
        for (int r = 0; r < PP; ++r) anyp |= pp[r].valid;
#ifdef SRL_NO_GROUND_ONLY
        const bool ground_only = false;
#else
        const bool ground_only = GB && __ballot(anyp) == 0ull;   // (wave-uniform)
#endif
        if (ground_only) {   // no pair point: the sweeps are the ground phase, the velocities stay in registers (ground_body_keep)
          v3 v = V(0.0f, 0.0f, 0.0f), w = V(0.0f, 0.0f, 0.0f);
          if (gb.np > 0) {
            const float4 a0 = *(const float4*)gb.pvw;
            const float2 a1 = *(const float2*)(gb.pvw + 4);
            v = V(a0.x, a0.z, a1.x); w = V(a0.y, a0.w, a1.y);
            float res = 0.0f;
            ground_body_keep<true, false>(L, gb, v, w, res);
          }
          for (int it = 0; it < P.c.solver_iterations; ++it) {
            done++;
            float res = 0.0f;
            if (gb.np > 0) ground_body_keep<false, GA>(L, gb, v, w, res);
            if (__ballot(res * res > P.c.residual_threshold) == 0ull) break;
          }
          if (gb.np > 0) { *(float4*)gb.pvw = make_float4(v.x, w.x, v.y, w.y); *(float2*)(gb.pvw + 4) = make_float2(v.z, w.z); }
        } else {
          solver_sweep<true, PP, PP == 1, GB, GA>(L, gp, gb, pp, ncol, gslot, pslot, 0, cm);
          for (int it = 0; it < P.c.solver_iterations; ++it) {
            done++;
            if (!solver_sweep<false, PP, PP == 1, GB, GA>(L, gp, gb, pp, ncol, gslot, pslot, 0, cm)) break;
          }
        }
        if (tid == 0) misc[M_CNT] = done;   // (M_CNT is free between the calls of newest_contacts)
      }
      __syncthreads();
      gsweep += misc[M_CNT];
    } else {
      solver_sweep<true, PP, false, GB, GA>(L, gp, gb, pp, ncol, gslot, pslot, 0, cm);
      for (int it = 0; it < P.c.solver_iterations; ++it) {
        gsweep++;
        if (!solver_sweep<false, PP, false, GB, GA>(L, gp, gb, pp, ncol, gslot, pslot, gsweep, cm)) break;
      }
    }
#ifdef SRL_STAMPS
    if (tid == 0) {
      EnvHdr* hh = &L.P->hdr[blockIdx.x];
      bool anyp = false;
      for (int r = 0; r < PP; ++r) anyp |= pp[r].valid;
      {   // pair turns of a sweep as scheduled (sum over colours of the fullest manifold) and as the longest dependency chain
        int bl[SRL_MAX_BODIES]; for (int b = 0; b < SRL_MAX_BODIES; ++b) bl[b] = 0;
        int lmax = 0, sumt = 0;
        for (int c = 0; c < ncol; ++c) {
          int mx = 0;
          for (int sl = 0; sl < P.NS; ++sl) {
            const int pid = L.POS()[sl];
            if (pid < 0 || L.COL()[sl] != c) continue;
            const int np = __float_as_int(L.MAN(sl)[0]);
            if (np <= 0) continue;
            int a, b; L.pair(pid, a, b);
            const int st = bl[a] > bl[b] ? bl[a] : bl[b];
            bl[a] = st + np; bl[b] = st + np;
            if (st + np > lmax) lmax = st + np;
            if (np > mx) mx = np;
          }
          sumt += mx;
        }
        hh->diag[4] += sumt; hh->diag[5] += lmax;
      }
      hh->diag[0] += 1; hh->diag[1] += (__ballot(anyp) == 0ull); hh->diag[2] += ncol; hh->diag[3] += solo ? misc[M_CNT] : 0;
    }
#endif
    // accumulated impulses back to the manifolds (warm start of the next sub-step)
    if (!GB) {
      if (gp.valid) { float* g = L.GM(gp.a); g[SRL_GM_IN + gp.idx] = gp.in; g[SRL_GM_T1 + gp.idx] = gp.i1; g[SRL_GM_T2 + gp.idx] = gp.i2; }
    } else if (gb.np > 0) {
      float* g = L.GM(gb.b);
#pragma unroll
      for (int i = 0; i < SRL_GMAXP; ++i)
        if (i < gb.np) { g[SRL_GM_IN + i] = gb.acc[i][0]; g[SRL_GM_T1 + i] = gb.acc[i][1]; g[SRL_GM_T2 + i] = gb.acc[i][2]; }
    }
#pragma unroll
    for (int r = 0; r < PP; ++r)
      if (pp[r].valid) {
        float* q = L.MAN((tid + r * T) >> 2) + 4 + SRL_MP_WORDS * pp[r].idx;
        q[10] = pp[r].in; q[11] = pp[r].i1; q[12] = pp[r].i2;
      }
  }
  STAMP(6);
  // (7) integrate (the last solver phase ended with a barrier)
  const float dt = P.c.sim_time_step;
  if (tid < nb) {
    const int b = tid;
    v3 v, w;
    ldvw(L, b, v, w);
    st3(L.X(b), madd(ld3(L.X(b)), v, dt));
    float* Q = L.Q(b);
    q4 q; q.x = Q[0]; q.y = Q[1]; q.z = Q[2]; q.w = Q[3];
    float hx = 0.5f * dt;
    q4 dq;
    dq.x = hx * ((w.x * q.w + w.y * q.z) - w.z * q.y);
    dq.y = hx * ((w.y * q.w + w.z * q.x) - w.x * q.z);
    dq.z = hx * ((w.z * q.w + w.x * q.y) - w.y * q.x);
    dq.w = hx * (-((w.x * q.x + w.y * q.y) + w.z * q.z));
    q.x += dq.x; q.y += dq.y; q.z += dq.z; q.w += dq.w;
    float inv = 1.0f / sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
    Q[0] = q.x * inv; Q[1] = q.y * inv; Q[2] = q.z * inv; Q[3] = q.w * inv;
  }
  __syncthreads();
  STAMP(7);
}

// simulator.py:322-335: every body's linear speed <= threshold
__device__ __forceinline__ bool sim_stop(const Lds& L, int nb, int tid) {
  int* misc = L.MISC();
  bool moving = false;
  if (tid < nb) {
    v3 v = ldv(L, tid);
    moving = sqrtf(dot(v, v)) > L.P->c.velocity_threshold;
  }
  // the bodies are lanes of wave 0 (nb <= 32): one ballot, one LDS word, one barrier.  (__syncthreads_or reads the
  // workgroup size from the dispatch packet in global memory on every call: a memory round trip per sub-step that a
  // lone wave cannot hide.)
  if (tid < 64) {
    const unsigned long long m = __ballot(moving);
    if (tid == 0) misc[M_MOVING] = m != 0ull;
  }
  __syncthreads();
  const int any = misc[M_MOVING];
  __syncthreads();   // the word is rewritten by the next call
  return !any;
}

// number of manifold points on the newest body (getContactPoints, simulator.py:340)
// (forced inline: called out of line, the Lds object — the layout offsets pinned in scalar registers — has to live in memory)
__device__ __forceinline__ int newest_contacts(const Lds& L, int nb, int tid, int T) {
  int* misc = L.MISC();
  if (tid == 0) misc[M_CNT] = __float_as_int(L.GM(nb - 1)[0]);
  __syncthreads();
  int n = 0;
  for (int sl = tid; sl < L.P->NS; sl += T) {
    int pid = L.POS()[sl];
    int pi_ = -1, pj_ = -1;
    if (pid >= 0) L.pair(pid, pi_, pj_);
    if (pid >= 0 && (pi_ == nb - 1 || pj_ == nb - 1)) n += __float_as_int(L.MAN(sl)[0]);
  }
  if (n) atomicAdd(&misc[M_CNT], n);
  __syncthreads();
  int r = misc[M_CNT];
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------ the rocks' render records, made in the env's own workgroup
// What srl_k_render's ray cast needs of every rock of this env at its final pose (stage.h).  The env is done; of a launch's
// workgroups all but the slowest reach this point long before the launch ends, so the staging rides on SIMDs that would idle
// (until round 5 it was a kernel of its own between the settle and the render kernel: 13.3 us per step).  A wave takes the
// rocks wave, wave + T / 64, ...; its LDS area lies in the scratch region behind the blob, which is dead after the last
// sub-step (the misc / pair words behind it are not touched: stackrl_hip.hip layout()); the next rock's geometry is
// requested before the current rock is computed.  Poses and mesh ids are read from the blob's LDS copy.
// NOT inlined: the sub-step loop's register allocation stays what it is without this code (inlined, the four variants'
// allocations moved: +8 VGPRs / +36 spilled registers and a private segment in kernels that had none).
__device__ __forceinline__ MeshHdr stage_mesh_hdr(const MeshHdr* __restrict__ mh, int m) {
  MeshHdr r;            // the six words the staging reads, through scalar registers (m is wave-uniform)
  r.vo = __builtin_amdgcn_readfirstlane(mh[m].vo); r.nv = __builtin_amdgcn_readfirstlane(mh[m].nv);
  r.to = __builtin_amdgcn_readfirstlane(mh[m].to); r.nt = __builtin_amdgcn_readfirstlane(mh[m].nt);
  r.eo = __builtin_amdgcn_readfirstlane(mh[m].eo); r.ne = __builtin_amdgcn_readfirstlane(mh[m].ne);
  r.inv_mass = 0.0f; r.iix = 0.0f; r.iiy = 0.0f; r.iiz = 0.0f; r.radius = 0.0f; r.cx = 0.0f; r.cy = 0.0f; r.cz = 0.0f;
  return r;
}

#ifdef SRL_STAGE_TAIL_INLINE    // (A / B builds only)
#define SRL_TAIL_ATTR __forceinline__
#else
#define SRL_TAIL_ATTR __attribute__((noinline))
#endif
template <int T>
__device__ SRL_TAIL_ATTR void stage_tail(const DevParams* __restrict__ Pp, float4* __restrict__ stage, int e, int nb) {
  const DevParams& P = *Pp;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  StageLds& S = ((StageLds*)(sm + P.BLOB))[wave];
  StageArgs A;
  A.mp = P.mp; A.mt = P.mt; A.me = P.me; A.mv = P.mv; A.px = P.px; A.inv_px = P.inv_px; A.res = P.c.overhead_res;
  const int slots = P.c.episode_length;
  const float* X = sm + P.OFF_X;
  const float* Q = sm + P.OFF_Q;
  const int* MESH = (const int*)(sm + P.OFF_MESH);
  float4* rec0 = stage + (size_t)e * slots * SRL_STAGE_STRIDE;
  int b = wave;
  if (b >= nb) return;
  MeshHdr mh = stage_mesh_hdr(P.mh, __builtin_amdgcn_readfirstlane(MESH[b]));
  StageLoads g = stage_request(A, mh, lane);
  for (;;) {
    const int bn = b + T / 64;
    const bool more = bn < nb;
    MeshHdr mh_n = mh;
    StageLoads g_n = g;
    if (more) { mh_n = stage_mesh_hdr(P.mh, __builtin_amdgcn_readfirstlane(MESH[bn])); g_n = stage_request(A, mh_n, lane); }
    const float* qq = Q + 4 * b;
    q4 q; q.x = qq[0]; q.y = qq[1]; q.z = qq[2]; q.w = qq[3];
    stage_compute(A, S, rec0 + (size_t)b * SRL_STAGE_STRIDE, ld3(X + 4 * b), q, mh, g, lane);
    __builtin_amdgcn_wave_barrier();
    if (!more) break;
    mh = mh_n; g = g_n; b = bn;
  }
}

// ------------------------------------------------------------------ K1 + K4 + episode machine
// Pp points to the handle's DevParams in device memory: every field access is a scalar load.  (A by-value
// kernel argument whose address is taken is copied to scratch and every access becomes a scratch load.)
template <int T, int PP>
__device__ __forceinline__ void step_body(const DevParams* __restrict__ Pp, const int64_t* __restrict__ action,
                                          int force_reset, const int32_t* __restrict__ order, float4* __restrict__ stage) {
  const DevParams& P = *Pp;
  extern __shared__ __attribute__((aligned(16))) float sm[];
#ifdef SRL_STEP_PRIO
  __builtin_amdgcn_s_setprio(SRL_STEP_PRIO);
#endif
  // order (may be NULL): the env this workgroup serves — a permutation of the batch written by srl_k_order_* below, envs
  // with the longest expected settle first.  Envs are independent, so the results do not depend on it.
  const int e = order ? order[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
  Lds L; L.init(sm, Pp);
  int* misc = L.MISC();
  EnvHdr* h = &P.hdr[e];
  float* gblob = P.blob + (size_t)e * P.BLOB;
#ifdef SRL_STAMPS
  if ((threadIdx.x & 63) == 0 && threadIdx.x < 128)   // which CU / SIMD the env's waves run on (HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20)
    h->hwid[threadIdx.x >> 6] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif

  if (tid == 0) {
    int mode;
    if (force_reset < 0) {                    // srl_step_simulation: -force_reset raw sub-steps, no placement
      mode = h->nb > 0 ? 4 : 3;
      misc[M_NB] = h->nb;
      misc[M_NCOL] = h->ncolour;
      misc[M_STATUS] = h->status;
      misc[M_RES0] = -1; misc[M_RES1] = -1;
    } else if (!force_reset && action[e] == (int64_t)SRL_ACTION_HOLD) {
      mode = 3;                               // this env sits the call out (srl_types.h)
    } else if (force_reset || h->done) {     // env.py:235-236 auto-reset
      env_reset(P, h, e);
      mode = 1;
    } else {
      int64_t a = action[e];
      int oi = 0, slot = 0;                   // TestStackEnv: action = (observation index, pixel), env.py:485-494
      if (P.n_slots > 1 && a >= 0) {
        // the index addresses the object maps on show: n_orient of the pending rock, or n_orient of every unplaced rock
        const int nvalid = P.c.ordering_freedom ? h->list_pos * P.n_orient : P.n_orient;
        slot = (int)(a / (int64_t)P.A);
        a = slot < nvalid ? a % (int64_t)P.A : -1;
        oi = slot % P.n_orient;
      }
      misc[M_ORIENT] = oi;
      if (a < 0 || a >= (int64_t)P.A) {       // env.py:238
        h->status |= SRL_ST_BAD_ACTION;
        atomicOr(P.flags, 1);
        mode = 2;
      } else {
        h->status &= ~SRL_ST_BAD_ACTION;
        mode = 0;
        misc[M_U] = (int)(a / P.AW);          // env.py:240-241
        misc[M_V] = (int)(a % P.AW);
        int next = -1, placed = h->pending;
        if (P.c.ordering_freedom) {           // TestSimulator.step: pop the chosen rock (simulator.py:372-378)
          const int rk = slot / P.n_orient;
          placed = h->ids[rk];
          for (int k = rk; k + 1 < h->list_pos; ++k) h->ids[k] = h->ids[k + 1];
          h->list_pos -= 1;
          if (h->list_pos == 0) h->done = 1;  // env.py:513-514: no objects left
          else next = h->ids[0];
        } else if (h->list_pos < P.c.episode_length) next = h->ids[h->list_pos++];   // env.py:243-247
        else h->done = 1;
        misc[M_NEXT] = next;
        misc[M_NB] = h->nb;
        misc[M_PENDING] = placed;
        misc[M_NCOL] = h->ncolour;
        misc[M_STATUS] = h->status;
        misc[M_ZMAX] = (int)f2o(-1e30f);
        misc[M_RES0] = -1; misc[M_RES1] = -1;
      }
    }
    h->mode = mode;
    misc[M_MODE] = mode;
  }
  __syncthreads();
  const int mode = misc[M_MODE];
  if (mode == 1) {   // Simulator.reset: empty world (simulator.py:156-188)
    int* gi = (int*)gblob;
    for (int k = tid; k < P.NP; k += T) gi[P.OFF_SOP + k] = -1;
    for (int k = tid; k < P.NS; k += T) gi[P.OFF_POS + k] = -1;
    return;
  }
  if (mode == 2 || mode == 3) return;

  // ---- load the persistent blob into LDS
  for (int k = tid; k < P.BLOB; k += T) sm[k] = gblob[k];
  for (int k = tid; k < P.NP; k += T) L.PAIR()[k] = pair_word(k);
  __syncthreads();

  // ---- K4: Observer.pose (observer.py:392-421): z = max(H[window] + O | O > 1e-4) - oz/2
  const int u = misc[M_U], v = misc[M_V], pending = misc[M_PENDING];
  int nb = misc[M_NB];
  if (mode == 0) {
    const int res = P.c.overhead_res, r = P.c.object_res;
    const float* Hm = P.H + (size_t)e * res * res;
    const float* Om = P.objmap + ((size_t)pending * P.n_orient + misc[M_ORIENT]) * r * r;
    uint32_t best = f2o(-1e30f);
    for (int k = tid; k < r * r; k += T) {
      int i = k / r, j = k % r;
      float o = Om[k];
      if (o > 1e-4f) {
        uint32_t s = f2o(Hm[(u + i) * res + (v + j)] + o);
        best = s > best ? s : best;
      }
    }
    atomicMax((uint32_t*)&misc[M_ZMAX], best);
  }
  __syncthreads();
  if (tid == 0 && mode == 0) {   // _place (simulator.py:310-320): teleport the pending rock, zero velocity
    float z = o2f((uint32_t)misc[M_ZMAX]);
    float half = ((float)P.c.object_res * P.px) * 0.5f;
    v3 pos = V((float)u * P.px + half, (float)v * P.px + half, z - half);
    const MeshHdr mh = P.mh[pending];
    int b = nb;
    L.MESH()[b] = pending;
    // resetBasePositionAndOrientation moves the inertial (COM) frame; loadURDF had placed the link frame
    if (P.n_orient == 1) {
      st3(L.X(b), P.c.place_at_com ? pos : pos + V(mh.cx, mh.cy, mh.cz));
      L.Q(b)[0] = 0.0f; L.Q(b)[1] = 0.0f; L.Q(b)[2] = 0.0f; L.Q(b)[3] = 1.0f;
    } else {   // the chosen orientation (observer.py:416-417 -> simulator.py:313)
      const float* oq = P.orient_q[misc[M_ORIENT]];
      q4 q; q.x = oq[0]; q.y = oq[1]; q.z = oq[2]; q.w = oq[3];
      st3(L.X(b), P.c.place_at_com ? pos : pos + mmul(quat_to_mat(q), V(mh.cx, mh.cy, mh.cz)));
      L.Q(b)[0] = q.x; L.Q(b)[1] = q.y; L.Q(b)[2] = q.z; L.Q(b)[3] = q.w;
    }
    stvw(L, b, V(0, 0, 0), V(0, 0, 0));
    L.GM(b)[0] = __int_as_float(0);
  }
  if (mode == 0) nb += 1;
  __syncthreads();
  for (int b = tid; b < nb; b += T) {
    int m = L.MESH()[b];
    const MeshHdr mh = P.mh[m];
    float* bc = L.BC(b);
    bc[0] = mh.inv_mass; bc[1] = mh.iix; bc[2] = mh.iiy; bc[3] = mh.iiz; bc[4] = mh.radius;
    bc[5] = __int_as_float(mh.nv); bc[6] = __int_as_float(mh.vo); bc[7] = __int_as_float(m);
  }
  __syncthreads();
  // local (COM-frame) vertices of every body into LDS, once per step (the 32-rock variant reads them from the mesh table
  // instead: without this copy two of its workgroups fit a CU)
  if (P.S_LV >= 0)
  for (int it = tid; it < nb * P.VS; it += T) {
    const int b = it / P.VS, k = it - b * P.VS;
    const float* bc = L.BC(b);
    if (k < __float_as_int(bc[5])) {
      float4 lv = P.mv[__float_as_int(bc[6]) + k];
      st3(L.LV(b) + 3 * k, V(lv.x, lv.y, lv.z));
    }
  }
  __syncthreads();

  // ---- Simulator.step (simulator.py:190-258) as one loop around a single sub-step call site:
  //   PLACE  : the sub-step of _place (simulator.py:320)
  //   SMOOTH : "zero the newest body's velocity, step" until _drop (simulator.py:212-224)
  //   SETTLE : step until _stop (simulator.py:239-245)
  enum { PH_PLACE = 0, PH_SMOOTH = 1, PH_ENTER = 2, PH_SETTLE = 3 };
  int counter = 0, phase = PH_PLACE, s_a = 0;
  int gsweep = 0;   // solver sweeps of this launch (block-uniform); the residual words start at -1
  bool diverged = false;
  if (mode == 4) {   // stepSimulation x n (srl_step_simulation): no placement, no stop criterion
    for (int k = 0; k < -force_reset; ++k) substep<T, PP>(L, nb, tid, gsweep);
    for (int k = tid; k < P.BLOB; k += T) gblob[k] = sm[k];
    if (tid == 0) { h->ncolour = misc[M_NCOL]; h->status = misc[M_STATUS]; h->sweeps = gsweep; }
    return;
  }
  for (;;) {
    if (phase == PH_SMOOTH) {
      if (tid == 0) stvw(L, nb - 1, V(0, 0, 0), V(0, 0, 0));   // resetBaseVelocity
      __syncthreads();
    }
    substep<T, PP>(L, nb, tid, gsweep);
    counter++;
    if (phase != PH_PLACE && counter > P.max_substeps) diverged = true;
    if (phase == PH_PLACE) phase = P.c.smooth_placing ? PH_SMOOTH : PH_ENTER;
    if (phase == PH_SMOOTH) {
      bool leave = diverged;
      if (!leave) {
        const bool drop = newest_contacts(L, nb, tid, T) >= 3;   // _drop, simulator.py:337-341
        const bool stop = sim_stop(L, nb, tid);
        leave = drop || stop;
      }
      if (leave) phase = PH_ENTER;
    }
    if (phase == PH_ENTER) {   // simulator.py:227-230: pose where the rock was left, steps before the drop
      if (tid == 0 && !diverged) {   // (the reference raises out of the smooth-placing loop at the cap: no place pose then)
        st3(L.PX(nb - 1), ld3(L.X(nb - 1)));
        for (int k = 0; k < 4; ++k) L.PQ(nb - 1)[k] = L.Q(nb - 1)[k];
      }
      s_a = counter;
      phase = PH_SETTLE;
    }
    if (phase == PH_SETTLE && (diverged || sim_stop(L, nb, tid))) break;
  }
  __syncthreads();

  // ---- write back
  for (int k = tid; k < P.BLOB; k += T) gblob[k] = sm[k];
  if (tid == 0) {
    h->nb = nb;
    h->pending = misc[M_NEXT];   // _load, simulator.py:258
    h->ncolour = misc[M_NCOL];
    h->substeps[0] = s_a;
    h->substeps[1] = counter - s_a;
    h->sweeps = gsweep;
    int st = misc[M_STATUS] | (diverged ? SRL_ST_DIVERGED : 0);
    h->status = st;
    if (diverged) atomicOr(P.flags, 2);
  }

  // ---- the rocks' render records (stage.h), by a function of its own: see stage_tail
#ifndef SRL_NO_STAGE_TAIL
  if (stage != nullptr) stage_tail<T>(Pp, stage, e, nb);
#endif
}

// Variants: T threads per env, one contact point of the body-body manifolds per thread (4 NS <= T; NS = 28 / 64 / 128
// slots up to 8 / 16 / 32 rocks, stackrl_hip.hip nslots).  L <= 8 runs two waves per env so that four envs per CU (1,024
// envs per GPU) are resident together with up to 256 VGPRs per lane; up to 16 rocks four waves; above: srl_k_step_pp2.  (Round 1
// kept a slot for every pair up to 16 rocks and 192 above, which needed two points per thread and 256 VGPRs + scratch.)
// The 16-rock variant is built for three waves per SIMD (168 VGPRs, 62 spilled to scratch): its shapes (2,048 - 4,096
// envs x 16 rocks) are throughput-bound, and a third workgroup per CU is worth more than the spills cost — 60.7 -> 50.1 ms
// per launch at 4,096 envs; four waves per SIMD (128 VGPRs) spill 108 registers and lose: 80 ms.
extern "C" __global__ void __launch_bounds__(128, 2) srl_k_step(const DevParams* __restrict__ Pp,
    const int64_t* __restrict__ action, int force_reset, const int32_t* __restrict__ order, float4* __restrict__ stage) {
  step_body<128, 1>(Pp, action, force_reset, order, stage);
}
extern "C" __global__ void __launch_bounds__(256, 3) srl_k_step_pp1(const DevParams* __restrict__ Pp,
    const int64_t* __restrict__ action, int force_reset, const int32_t* __restrict__ order, float4* __restrict__ stage) {
  step_body<256, 1>(Pp, action, force_reset, order, stage);
}
// Above 16 rocks: four waves with two points per thread (128 slots) and no LDS copy of the local vertices — 70 KB per env,
// two workgroups per CU.  (Eight waves with one point per thread and the vertex copy, 97 KB and one workgroup per CU:
// 145.5 against 104.9 ms per launch at 2,048 envs x 32 rocks.)
extern "C" __global__ void __launch_bounds__(256, 2) srl_k_step_pp2(const DevParams* __restrict__ Pp,
    const int64_t* __restrict__ action, int force_reset, const int32_t* __restrict__ order, float4* __restrict__ stage) {
  step_body<256, 2>(Pp, action, force_reset, order, stage);
}

// 9 - 16 rocks, large batches: two waves per env with two points per thread and no LDS copy of the local vertices —
// 35 KB per env, four workgroups per CU instead of three (the shapes with >= 2,048 envs are throughput-bound)
extern "C" __global__ void __launch_bounds__(128, 2) srl_k_step_t128(const DevParams* __restrict__ Pp,
    const int64_t* __restrict__ action, int force_reset, const int32_t* __restrict__ order, float4* __restrict__ stage) {
  step_body<128, 2>(Pp, action, force_reset, order, stage);
}

// ------------------------------------------------------------------ launch order of a batch that outnumbers the resident slots
// A settle launch lasts as long as its slowest env (stop criterion simulator.py:322-335) and the dispatcher hands workgroups
// out in index order: with more envs than resident workgroups (4,096 x 16 rocks: four rounds) a long chain that starts in
// the last round sets the end of the launch.  What can be known before the launch is where the rock is released:
// the height `Observer.pose` will compute (observer.py:405-413: max of H[window] + O over the rock's pixels) — the higher the
// release, the more rocks lie underneath and the longer the settle (Spearman 0.5 - 0.75 against the oracle's sweep counts,
// tests/diag/sched_predictors.py).  srl_k_order_keys evaluates it per env (one wave each), srl_k_order_sort sorts the batch
// by it (one workgroup, bitonic network in LDS), highest first; srl_k_step then serves env order[blockIdx.x].
extern "C" __global__ void __launch_bounds__(64) srl_k_order_keys(const DevParams* __restrict__ Pp,
    const int64_t* __restrict__ action, unsigned long long* __restrict__ keys) {
  const DevParams& P = *Pp;
  const int e = blockIdx.x, lane = threadIdx.x;
  const EnvHdr* h = &P.hdr[e];
  float z = -1.0f;                                   // auto-reset calls, held envs, rejected actions: no settle, last
  int64_t a = action[e];
  if (a != (int64_t)SRL_ACTION_HOLD && !h->done && a >= 0) {
    int slot = 0, mesh = h->pending, nvalid = 1;
    if (P.n_slots > 1) {
      nvalid = P.c.ordering_freedom ? h->list_pos * P.n_orient : P.n_orient;
      slot = (int)(a / (int64_t)P.A);
      a = a % (int64_t)P.A;
      if (P.c.ordering_freedom && slot < nvalid) mesh = h->ids[slot / P.n_orient];
    }
    if (slot < nvalid && a < (int64_t)P.A && mesh >= 0) {
      const int res = P.c.overhead_res, r = P.c.object_res;
      const int u = (int)(a / P.AW), v = (int)(a % P.AW);
      const float* Hm = P.H + (size_t)e * res * res;
      const float* Om = P.objmap + ((size_t)mesh * P.n_orient + slot % P.n_orient) * r * r;
      float best = 0.0f;
      for (int k = lane; k < r * r; k += 64) {
        const float o = Om[k];
        if (o > 1e-4f) best = fmaxf(best, Hm[(u + k / r) * res + (v + k % r)] + o);
      }
      for (int s = 32; s >= 1; s >>= 1) best = fmaxf(best, __shfl_xor(best, s));
      z = best;
    }
  }
  // ascending sort -> highest release first, ties in index order; non-negative floats order like their bit patterns
  if (lane == 0) keys[e] = ((unsigned long long)(z < 0.0f ? 0xffffffffu : ~__float_as_uint(z)) << 32) | (unsigned)e;
}

extern "C" __global__ void __launch_bounds__(1024) srl_k_order_sort(const unsigned long long* __restrict__ keys, int n, int np2,
                                                                    int32_t* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const int tid = threadIdx.x;
  for (int i = tid; i < np2; i += 1024) sk[i] = i < n ? keys[i] : ~0ull;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j >= 1; j >>= 1) {
      for (int t = tid; t < (np2 >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;      // the pair (i, i + j) of this stage
        const unsigned long long a = sk[i], b = sk[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { sk[i] = b; sk[p] = a; }
      }
      __syncthreads();
    }
  for (int i = tid; i < n; i += 1024) order[i] = (int32_t)(unsigned)(sk[i] & 0xffffffffull);
}

// ------------------------------------------------------------------ ParallelEnv.sample (utils.py:534-538)
extern "C" __global__ void srl_k_sample(DevParams P, int64_t* __restrict__ action) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.c.n_envs) return;
  uint32_t key = P.seed + (uint32_t)P.c.env_index_offset + (uint32_t)i;
  int nvalid = P.n_orient;   // object maps on show: with ordering freedom those of the rocks still unplaced
  if (P.c.ordering_freedom) { const int left = P.hdr[i].list_pos; nvalid *= left > 0 ? left : 1; }
  action[i] = (int64_t)srl_rng_below(srl_rng(key, P.sample_counter, SRL_STREAM_ACTION, 0), (uint32_t)(P.A * nvalid));
}

// ------------------------------------------------------------------ telemetry reduction
extern "C" __global__ void srl_k_contacts(DevParams P, float* __restrict__ max_pen, int32_t* __restrict__ n_points) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.c.n_envs) return;
  const float* gb = P.blob + (size_t)e * P.BLOB;
  const int* gi = (const int*)gb;
  int nb = P.hdr[e].nb;
  float mp = 0.0f; int np = 0;
  for (int b = 0; b < nb; ++b) {
    const float* g = gb + P.OFF_GM + SRL_GM_WORDS * b;
    int n = __float_as_int(g[0]);
    for (int k = 0; k < n; ++k) { np++; if (-g[SRL_GM_DIST + k] > mp) mp = -g[SRL_GM_DIST + k]; }
  }
  for (int sl = 0; sl < P.NS; ++sl) {
    if (gi[P.OFF_POS + sl] < 0) continue;
    const float* m = gb + P.OFF_MAN + SRL_MAN_WORDS * sl;
    int n = __float_as_int(m[0]);
    for (int k = 0; k < n; ++k) { np++; float d = m[4 + SRL_MP_WORDS * k + 9]; if (-d > mp) mp = -d; }
  }
  max_pen[e] = mp; n_points[e] = np;
}
