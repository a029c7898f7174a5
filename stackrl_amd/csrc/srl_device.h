// srl_device.h — device-side building blocks of libstackrl_hip (gfx950 only).
//
// All arithmetic is IEEE binary32 with one rounding per written operation (the library is built
// with -ffp-contract=off): the kernels evaluate exactly the expression trees of the solver /
// rasteriser definition in DESIGN.md, so results do not depend on how work is spread over lanes.
//
// Reference rows implemented here (paths relative to menezesandre/stackrl):
//   observer.py:259-260, :274-277   depth -> elevation (elev_overhead / elev_object)
//   rewarder.py:225-259             goal rectangle (goal_from_rng)
//   simulator.py:190-341            place / smooth placing / settle (settle.hip uses these pieces)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SRL_FAR 1000.0f      // Observer.far, observer.py:6
#define SRL_GMAXP 4          // ground manifold points per body: Bullet's MANIFOLD_CACHE_SIZE (round 1 kept 8)
#define SRL_NSLOT_MAX 192    // persistent body-body manifolds per env
#define SRL_MP_WORDS 13      // manifold point: la3 lb3 n3 dist in it1 it2
#define SRL_MAN_WORDS 60     // np, axis3, 4 points, cached GJK simplex: n, 3 words of packed (ia, ib) pairs
#define SRL_GM_WORDS (1 + 5 * SRL_GMAXP)   // np, vid[], dist[], in[], it1[], it2[]
#define SRL_GM_VID 1
#define SRL_GM_DIST (1 + SRL_GMAXP)
#define SRL_GM_IN (1 + 2 * SRL_GMAXP)
#define SRL_GM_T1 (1 + 3 * SRL_GMAXP)
#define SRL_GM_T2 (1 + 4 * SRL_GMAXP)
#define SRL_GJK_MAXIT 32

struct v3 { float x, y, z; };
struct q4 { float x, y, z, w; };
struct m3 { float m[9]; };

__device__ __forceinline__ v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
// The vector kernels are DEFINED with fused multiply-adds (one rounding per fmaf), mirrored by the oracle.
__device__ __forceinline__ float dot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
  return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ v3 madd(v3 a, v3 b, float s) {   // a + b * s
  return V(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z));
}
__device__ __forceinline__ v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ m3 ldm(const float* p) {
  m3 R;
#pragma unroll
  for (int i = 0; i < 9; ++i) R.m[i] = p[i];
  return R;
}
// Diagnostic builds: -DSRL_BISECT=<bits> compiles single helpers without any optimisation, i.e.
// without the SLP vectoriser's packed code, to find which one a concurrency-dependent result comes from.
#ifndef SRL_BISECT
#define SRL_BISECT 0
#endif
#define SRL_HELPER(bit) __device__ __attribute__((noinline, optnone))
#if SRL_BISECT & 2
SRL_HELPER(1) v3 mmul(const m3& R, v3 a) {
#else
__device__ __forceinline__ v3 mmul(const m3& R, v3 a) {
#endif
  return V(fmaf(R.m[0], a.x, fmaf(R.m[1], a.y, R.m[2] * a.z)), fmaf(R.m[3], a.x, fmaf(R.m[4], a.y, R.m[5] * a.z)),
           fmaf(R.m[6], a.x, fmaf(R.m[7], a.y, R.m[8] * a.z)));
}
#if SRL_BISECT & 2
SRL_HELPER(1) v3 mmul_add(const m3& R, v3 a, v3 x) {   // x + R * a
#else
__device__ __forceinline__ v3 mmul_add(const m3& R, v3 a, v3 x) {   // x + R * a
#endif
  return V(fmaf(R.m[0], a.x, fmaf(R.m[1], a.y, fmaf(R.m[2], a.z, x.x))),
           fmaf(R.m[3], a.x, fmaf(R.m[4], a.y, fmaf(R.m[5], a.z, x.y))),
           fmaf(R.m[6], a.x, fmaf(R.m[7], a.y, fmaf(R.m[8], a.z, x.z))));
}
__device__ __forceinline__ v3 mtmul(const m3& R, v3 a) {
  return V(fmaf(R.m[0], a.x, fmaf(R.m[3], a.y, R.m[6] * a.z)), fmaf(R.m[1], a.x, fmaf(R.m[4], a.y, R.m[7] * a.z)),
           fmaf(R.m[2], a.x, fmaf(R.m[5], a.y, R.m[8] * a.z)));
}
#if SRL_BISECT & 1
SRL_HELPER(0) m3 quat_to_mat(q4 q) {
#else
__device__ __forceinline__ m3 quat_to_mat(q4 q) {
#endif
  float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
  float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
  float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  m3 R;
  R.m[0] = 1.0f - 2.0f * (yy + zz); R.m[1] = 2.0f * (xy - wz); R.m[2] = 2.0f * (xz + wy);
  R.m[3] = 2.0f * (xy + wz); R.m[4] = 1.0f - 2.0f * (xx + zz); R.m[5] = 2.0f * (yz - wx);
  R.m[6] = 2.0f * (xz - wy); R.m[7] = 2.0f * (yz + wx); R.m[8] = 1.0f - 2.0f * (xx + yy);
  return R;
}
// world inverse inertia R diag(d) R^T
__device__ __forceinline__ m3 inv_inertia_world(const m3& R, v3 d) {
  m3 A;
  A.m[0] = R.m[0] * d.x; A.m[1] = R.m[1] * d.y; A.m[2] = R.m[2] * d.z;
  A.m[3] = R.m[3] * d.x; A.m[4] = R.m[4] * d.y; A.m[5] = R.m[5] * d.z;
  A.m[6] = R.m[6] * d.x; A.m[7] = R.m[7] * d.y; A.m[8] = R.m[8] * d.z;
  m3 I;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      I.m[3 * i + j] = (A.m[3 * i] * R.m[3 * j] + A.m[3 * i + 1] * R.m[3 * j + 1]) + A.m[3 * i + 2] * R.m[3 * j + 2];
  return I;
}
// Bullet btPlaneSpace1 restated
__device__ __forceinline__ void plane_space(v3 n, v3& p, v3& q) {
  if (fabsf(n.z) > 0.70710678f) {
    float a = n.y * n.y + n.z * n.z;
    float k = 1.0f / sqrtf(a);
    p = V(0.0f, -n.z * k, n.y * k);
    q = V(a * k, -n.x * p.z, n.x * p.y);
  } else {
    float a = n.x * n.x + n.y * n.y;
    float k = 1.0f / sqrtf(a);
    p = V(-n.y * k, n.x * k, 0.0f);
    q = V(-n.z * p.y, n.z * p.x, a * k);
  }
}
// private acos polynomial (A&S 4.4.46): bit-identical on host and device
__device__ __forceinline__ float srl_acosf(float x) {
  float a = fabsf(x);
  if (a > 1.0f) a = 1.0f;
  float p = -0.0012624911f;
  p = p * a + 0.0066700901f;
  p = p * a + -0.0170881256f;
  p = p * a + 0.0308918810f;
  p = p * a + -0.0501743046f;
  p = p * a + 0.0889789874f;
  p = p * a + -0.2145988016f;
  p = p * a + 1.5707963050f;
  float r = sqrtf(1.0f - a) * p;
  return x < 0.0f ? 3.14159265358979f - r : r;
}

// ---------------------------------------------------------------- ordered-int view of floats
__device__ __forceinline__ uint32_t f2o(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float o2f(uint32_t o) {
  uint32_t u = o ^ (((o >> 31) - 1u) | 0x80000000u);
  return __uint_as_float(u);
}

// ---------------------------------------------------------------- counter RNG
__host__ __device__ __forceinline__ uint32_t srl_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t srl_rng(uint32_t key, uint32_t episode, uint32_t stream, uint32_t draw) {
  uint32_t h = srl_mix32(key + 0x9E3779B9U);
  h = srl_mix32(h ^ (episode + 0x85EBCA6BU));
  h = srl_mix32(h ^ (stream + 0xC2B2AE35U));
  h = srl_mix32(h ^ (draw + 0x27D4EB2FU));
  return h;
}
__host__ __device__ __forceinline__ uint32_t srl_rng_below(uint32_t r, uint32_t n) {
  return (uint32_t)(((uint64_t)r * (uint64_t)n) >> 32);
}
enum { SRL_STREAM_MESH = 0, SRL_STREAM_GOAL = 1, SRL_STREAM_ACTION = 2 };

// ---------------------------------------------------------------- depth codec
// pybullet TinyRenderer depth restated: d = far (t - near) / (t (far - near))
__device__ __forceinline__ float depth_encode(float t, float nearp, float farp) {
  if (t < nearp) t = nearp;
  if (t > farp) t = farp;
  return (farp * (t - nearp)) / (t * (farp - nearp));
}

// ---------------------------------------------------------------- GJK (Ericson closest-point sub-algorithms)
__device__ __forceinline__ void closest_tri(v3 a, v3 b, v3 c, float& l0, float& l1, float& l2, int& used) {
  v3 ab = b - a, ac = c - a, ap = neg(a);
  float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) { l0 = 1.0f; l1 = 0.0f; l2 = 0.0f; used = 1; return; }
  v3 bp = neg(b);
  float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) { l0 = 0.0f; l1 = 1.0f; l2 = 0.0f; used = 2; return; }
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
    float v = d1 / (d1 - d3);
    l0 = 1.0f - v; l1 = v; l2 = 0.0f; used = 3; return;
  }
  v3 cp = neg(c);
  float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) { l0 = 0.0f; l1 = 0.0f; l2 = 1.0f; used = 4; return; }
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
    float w = d2 / (d2 - d6);
    l0 = 1.0f - w; l1 = 0.0f; l2 = w; used = 5; return;
  }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
    float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    l0 = 0.0f; l1 = 1.0f - w; l2 = w; used = 6; return;
  }
  float denom = 1.0f / ((va + vb) + vc);
  float v = vb * denom, w = vc * denom;
  l0 = (1.0f - v) - w; l1 = v; l2 = w; used = 7;
}

__device__ __forceinline__ int outside_plane(v3 a, v3 b, v3 c, v3 d) {
  v3 n = cross(b - a, c - a);
  float sp = dot(neg(a), n);
  float sd = dot(d - a, n);
  if (sd * sd < 1e-24f) return -1;
  return (sp * sd < 0.0f) ? 1 : 0;
}

struct Simplex {
  v3 w[4], p[4], q[4];
  int ia[4], ib[4];
  float lam[4];
  int n;
};

// all indexing below is static after unrolling (no scratch)
__device__ __forceinline__ void simplex_push(Simplex& s, v3 w, v3 p, v3 q, int ia, int ib) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (s.n == i) { s.w[i] = w; s.p[i] = p; s.q[i] = q; s.ia[i] = ia; s.ib[i] = ib; }
  s.n += 1;
}

// returns 1 ok (simplex reduced, lam + v valid), 0 degenerate (nothing modified), 2 origin enclosed
__device__ __forceinline__ int simplex_closest(Simplex& s, v3& vout) {
  int used = 0;
  float l[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (s.n == 1) {
    l[0] = 1.0f; used = 1;
  } else if (s.n == 2) {
    v3 a = s.w[0], b = s.w[1];
    v3 ab = b - a;
    float t = dot(neg(a), ab);
    if (t <= 0.0f) { l[0] = 1.0f; used = 1; }
    else {
      float den = dot(ab, ab);
      if (t >= den) { l[1] = 1.0f; used = 2; }
      else { t = t / den; l[0] = 1.0f - t; l[1] = t; used = 3; }
    }
  } else if (s.n == 3) {
    closest_tri(s.w[0], s.w[1], s.w[2], l[0], l[1], l[2], used);
  } else {
    v3 a = s.w[0], b = s.w[1], c = s.w[2], d = s.w[3];
    int o0 = outside_plane(a, b, c, d);
    int o1 = outside_plane(a, c, d, b);
    int o2 = outside_plane(a, d, b, c);
    int o3 = outside_plane(b, d, c, a);
    if (o0 < 0 || o1 < 0 || o2 < 0 || o3 < 0) return 0;
    if (!o0 && !o1 && !o2 && !o3) return 2;
    float best = 1e30f;
    float t0, t1, t2; int tu;
    if (o0) {
      closest_tri(a, b, c, t0, t1, t2, tu);
      v3 p = madd(madd(a * t0, b, t1), c, t2);
      float d2 = dot(p, p);
      if (d2 < best) { best = d2; l[0] = t0; l[1] = t1; l[2] = t2; l[3] = 0.0f; used = (tu & 1) | (tu & 2) | (tu & 4); }
    }
    if (o1) {
      closest_tri(a, c, d, t0, t1, t2, tu);
      v3 p = madd(madd(a * t0, c, t1), d, t2);
      float d2 = dot(p, p);
      if (d2 < best) { best = d2; l[0] = t0; l[1] = 0.0f; l[2] = t1; l[3] = t2; used = (tu & 1) | ((tu & 2) << 1) | ((tu & 4) << 1); }
    }
    if (o2) {
      closest_tri(a, d, b, t0, t1, t2, tu);
      v3 p = madd(madd(a * t0, d, t1), b, t2);
      float d2 = dot(p, p);
      if (d2 < best) { best = d2; l[0] = t0; l[1] = t2; l[2] = 0.0f; l[3] = t1; used = (tu & 1) | ((tu & 2) << 2) | ((tu & 4) >> 1); }
    }
    if (o3) {
      closest_tri(b, d, c, t0, t1, t2, tu);
      v3 p = madd(madd(b * t0, d, t1), c, t2);
      float d2 = dot(p, p);
      if (d2 < best) { best = d2; l[0] = 0.0f; l[1] = t0; l[2] = t2; l[3] = t1; used = ((tu & 1) << 1) | ((tu & 2) << 2) | (tu & 4); }
    }
  }
  // compaction, dropping unused vertices from the highest index down (keeps ascending order)
#define SRL_SHIFT(i)                                                                                 \
  if (!(used & (1 << (i)))) {                                                                        \
    _Pragma("unroll") for (int k = (i); k < 3; ++k) {                                                \
      s.w[k] = s.w[k + 1]; s.p[k] = s.p[k + 1]; s.q[k] = s.q[k + 1];                                  \
      s.ia[k] = s.ia[k + 1]; s.ib[k] = s.ib[k + 1]; l[k] = l[k + 1];                                  \
    }                                                                                                \
  }
  SRL_SHIFT(3) SRL_SHIFT(2) SRL_SHIFT(1) SRL_SHIFT(0)
#undef SRL_SHIFT
  int m = __popc((unsigned)used & ((1u << s.n) - 1u));
  v3 v = V(0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < m) { s.lam[k] = l[k]; v = madd(v, s.w[k], l[k]); }
  s.n = m;
  vout = v;
  return 1;
}

// Exchange steps of symmetric all-reductions over aligned groups of 16 / 32 lanes without LDS traffic (__shfl_xor
// compiles to ds_bpermute_b32: an LDS round trip per step, which a lone wave cannot hide).  Steps within a row of 16
// lanes are DPP moves — quad_perm [1,0,3,2] (0xB1), quad_perm [2,3,0,1] (0x4E), row_half_mirror (0x141), row_mirror
// (0x140): after the first two a quad is uniform, the mirrors then pair quad with quad and half with half — and the step
// between the two rows of a 32-lane group is v_permlane16_swap_b32 (gfx950), which hands every lane both rows' values at
// its position.  The pairing differs from xor-butterflies but a reduction by a total order gives the same result
// (tools/experiments/permlane_swap.hip prints the lane maps).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
__device__ __forceinline__ void rows_i(int v, int& even, int& odd) {
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  even = (int)r[0]; odd = (int)r[1];
}
__device__ __forceinline__ void rows_f(float v, float& even, float& odd) {
  int a, b; rows_i(__float_as_int(v), a, b);
  even = __int_as_float(a); odd = __int_as_float(b);
}

// 64-bit forms of the exchange steps, for reductions on (ordered float << 32 | index) keys: one 64-bit integer
// compare decides a lexicographic (value, index) order, which keeps the reduction's dependency chain at
// move -> compare -> select instead of three compares joined by scalar logic
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {
  return ((uint64_t)(uint32_t)dpp_i<CTRL>((int)(v >> 32)) << 32) | (uint32_t)dpp_i<CTRL>((int)(uint32_t)v);
}
__device__ __forceinline__ void rows_u64(uint64_t v, uint64_t& even, uint64_t& odd) {
  int h0, h1, l0, l1;
  rows_i((int)(v >> 32), h0, h1); rows_i((int)(uint32_t)v, l0, l1);
  even = ((uint64_t)(uint32_t)h0 << 32) | (uint32_t)l0; odd = ((uint64_t)(uint32_t)h1 << 32) | (uint32_t)l1;
}
// key of (value, index) for "largest value, lowest index on ties" (x + 0 turns -0 into +0: the float order has one zero)
__device__ __forceinline__ uint64_t key_max_lo(float d, int k) { return ((uint64_t)f2o(d + 0.0f) << 32) | (uint32_t)(0x7fffffff - k); }

// Support vertices of A in direction -v and of B in direction +v (arg max of P[k] . d, lowest index on ties),
// computed by a group of G adjacent lanes: each lane scans the vertices k = gl, gl + G, ... of both clouds and the
// group combines both results in the same xor-shuffle rounds.  G = 1 is the plain sequential scan.  The dot
// products are the same on every mapping, so the indices returned do not depend on G.
template <int G>
__device__ __forceinline__ void support_pair(const float* VA, int na, const float* VB, int nb, v3 v, int gl, int& ia,
                                             int& ib) {
  const v3 nv = neg(v);
  int ba = 0x7fffffff, bb = 0x7fffffff;
  float da = -3.0e38f, db = -3.0e38f;
  const int n = na > nb ? na : nb;
  for (int k = gl; k < n; k += G) {
    if (k < na) { float t = dot(ld3(VA + 3 * k), nv); if (t > da) { da = t; ba = k; } }
    if (k < nb) { float t = dot(ld3(VB + 3 * k), v); if (t > db) { db = t; bb = k; } }
  }
  static_assert(G == 1 || G == 16 || G == 32, "group reductions are written for 1, 16 or 32 lanes");
  if (G > 1) {
    // arg max of (dot, lowest index) over the group on packed keys; the sentinel index 0x7fffffff (no vertex seen) packs
    // to 0 and loses every tie, as in the sequential scan
    uint64_t ka = key_max_lo(da, ba), kb = key_max_lo(db, bb);
#define SRL_SUP_STEP(C) { const uint64_t oa = dpp_u64<C>(ka), ob = dpp_u64<C>(kb); ka = oa > ka ? oa : ka; kb = ob > kb ? ob : kb; }
    SRL_SUP_STEP(0xB1) SRL_SUP_STEP(0x4E) SRL_SUP_STEP(0x141) SRL_SUP_STEP(0x140)
#undef SRL_SUP_STEP
    if (G >= 32) {
      uint64_t a0, a1, b0, b1;
      rows_u64(ka, a0, a1); rows_u64(kb, b0, b1);
      ka = a1 > a0 ? a1 : a0; kb = b1 > b0 ? b1 : b0;
    }
    ba = 0x7fffffff - (int)(uint32_t)ka; bb = 0x7fffffff - (int)(uint32_t)kb;
  }
  ia = ba; ib = bb;
}

// GJK distance between two world-space vertex clouds held in LDS (float triples), run by a group of G lanes that
// all follow the same control flow (only the support scans are split).  `cache` (4 words: n, then the (ia, ib)
// pairs packed two per word) holds the simplex of the previous call: re-evaluated at the current poses it is the
// starting simplex, so a resting contact converges in one iteration; a degenerate or enclosing cached simplex
// restarts from the cached axis.
// 0: farther than maxdist; 1: pa/pb/n/dist valid; 2: hulls overlap.
template <int G>
__device__ __forceinline__ int gjk_distance(const float* VA, int na, const float* VB, int nb, v3& axis, int* cache,
                                            float maxdist, v3& pa, v3& pb, v3& nrm, float& dist, int gl) {
  Simplex s;
  s.n = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s.lam[k] = 0.0f; s.ia[k] = -1; s.ib[k] = -1;
    s.w[k] = V(0.0f, 0.0f, 0.0f); s.p[k] = V(0.0f, 0.0f, 0.0f); s.q[k] = V(0.0f, 0.0f, 0.0f);
  }
  v3 v = axis;
  float sqd = 1e30f;
  bool warm = false;
  const int cn = cache[0];
  if (cn > 0) {
    const int c1 = cache[1], c2 = cache[2];
    s.n = cn;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < cn) {
        const int pr = ((k < 2 ? c1 : c2) >> (16 * (k & 1))) & 0xffff;
        s.ia[k] = pr & 0xff; s.ib[k] = pr >> 8;
        s.p[k] = ld3(VA + 3 * s.ia[k]); s.q[k] = ld3(VB + 3 * s.ib[k]);
        s.w[k] = s.p[k] - s.q[k];
      }
    }
    v3 nv0;
    int rc = simplex_closest(s, nv0);
    float nsq = dot(nv0, nv0);
    if (rc == 1 && !(nsq < 1e-10f)) { v = nv0; sqd = nsq; warm = true; }
    else {
      s.n = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s.lam[k] = 0.0f;
    }
  }
  if (!(dot(v, v) > 1e-20f)) v = V(0.0f, 0.0f, 1.0f);
  int result = -1;
  for (int it = 0; it < SRL_GJK_MAXIT; ++it) {
    int ia, ib;
    support_pair<G>(VA, na, VB, nb, v, gl, ia, ib);
    v3 a = ld3(VA + 3 * ia), b = ld3(VB + 3 * ib);
    v3 w = a - b;
    float delta = dot(v, w);
    if (it > 0 || warm) {
      if (delta > 0.0f && delta * delta > sqd * (maxdist * maxdist)) { result = 0; break; }
      int dup = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) dup |= (k < s.n && s.ia[k] == ia && s.ib[k] == ib);
      if (dup) break;
      if (sqd - delta <= sqd * 1e-6f) break;
    }
    simplex_push(s, w, a, b, ia, ib);
    v3 nv;
    int rc = simplex_closest(s, nv);
    if (rc == 2) { cache[0] = 0; return 2; }
    if (rc == 0) {
      if (it == 0 && !warm) { cache[0] = 0; return 0; }
      s.n -= 1;   // drop the vertex just pushed; lam still describes the previous simplex
      break;
    }
    float nsq = dot(nv, nv);
    if (nsq < 1e-10f) { cache[0] = 0; return 2; }
    bool stall = (it > 0 || warm) && (sqd - nsq <= 1.1920929e-7f * sqd);
    v = nv; sqd = nsq;
    if (stall) break;
  }
  {   // store the simplex for the next call
    int pk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pk[k] = k < s.n ? ((s.ia[k] & 0xff) | ((s.ib[k] & 0xff) << 8)) : 0;
    cache[0] = s.n; cache[1] = pk[0] | (pk[1] << 16); cache[2] = pk[2] | (pk[3] << 16);
  }
  if (result == 0) return 0;
  v3 A = V(0.0f, 0.0f, 0.0f), B = V(0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < s.n) { A = madd(A, s.p[k], s.lam[k]); B = madd(B, s.q[k], s.lam[k]); }
  float d = sqrtf(sqd);
  if (d > maxdist) { axis = v; return 0; }
  pa = A; pb = B; dist = d;
  nrm = v * (1.0f / d);
  axis = v;
  return 1;
}
