/* srl_oracle.c — CPU oracle for the Stack-v0 hot path.  TEST INFRASTRUCTURE ONLY.
 * See srl_oracle.h for scope and the parity-pinning statement.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  All arithmetic is
 * IEEE binary32 with one rounding per written operation, so that an independent
 * implementation evaluating the same expressions (the HIP kernels) is bit-identical.
 *
 * Plain sequential C: one env at a time, one body / pair / pixel at a time.
 */
#include "srl_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXB SRL_MAX_BODIES
#define MAXV SRL_MAX_VERTS
#define MAXT SRL_MAX_TRIS
#define MAXSLOT 192               /* persistent body-body manifolds per env */
#define NPAIR (MAXB * (MAXB - 1) / 2)
#define GJK_MAXIT 32
#define FAR_PLANE 1000.0f         /* Observer.far, observer.py:6 */

static char g_err[256];
const char* srlo_last_error(void) { return g_err; }
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

/* ------------------------------------------------------------------ math */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } q4;
typedef struct { float m[9]; } m3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* The vector kernels below are DEFINED with fused multiply-adds (one rounding per fmaf): the HIP side
 * issues v_fma_f32 for exactly these expressions. */
static inline float vdot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline v3 vcross(v3 a, v3 b) {
  return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 vmadd(v3 a, v3 b, float s) { /* a + b * s */
  return V(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z));
}
static inline v3 mmul(const m3* R, v3 a) { /* R * a */
  return V(fmaf(R->m[0], a.x, fmaf(R->m[1], a.y, R->m[2] * a.z)),
           fmaf(R->m[3], a.x, fmaf(R->m[4], a.y, R->m[5] * a.z)),
           fmaf(R->m[6], a.x, fmaf(R->m[7], a.y, R->m[8] * a.z)));
}
static inline v3 mmul_add(const m3* R, v3 a, v3 x) { /* x + R * a */
  return V(fmaf(R->m[0], a.x, fmaf(R->m[1], a.y, fmaf(R->m[2], a.z, x.x))),
           fmaf(R->m[3], a.x, fmaf(R->m[4], a.y, fmaf(R->m[5], a.z, x.y))),
           fmaf(R->m[6], a.x, fmaf(R->m[7], a.y, fmaf(R->m[8], a.z, x.z))));
}
static inline v3 mtmul(const m3* R, v3 a) { /* R^T * a */
  return V(fmaf(R->m[0], a.x, fmaf(R->m[3], a.y, R->m[6] * a.z)),
           fmaf(R->m[1], a.x, fmaf(R->m[4], a.y, R->m[7] * a.z)),
           fmaf(R->m[2], a.x, fmaf(R->m[5], a.y, R->m[8] * a.z)));
}
static inline m3 quat_to_mat(q4 q) {
  float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
  float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
  float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  m3 R;
  R.m[0] = 1.0f - 2.0f * (yy + zz); R.m[1] = 2.0f * (xy - wz); R.m[2] = 2.0f * (xz + wy);
  R.m[3] = 2.0f * (xy + wz); R.m[4] = 1.0f - 2.0f * (xx + zz); R.m[5] = 2.0f * (yz - wx);
  R.m[6] = 2.0f * (xz - wy); R.m[7] = 2.0f * (yz + wx); R.m[8] = 1.0f - 2.0f * (xx + yy);
  return R;
}
/* world inverse inertia  R diag(d) R^T  (symmetric, stored full) */
static inline m3 inv_inertia_world(const m3* R, v3 d) {
  m3 A; /* A = R diag(d) */
  A.m[0] = R->m[0] * d.x; A.m[1] = R->m[1] * d.y; A.m[2] = R->m[2] * d.z;
  A.m[3] = R->m[3] * d.x; A.m[4] = R->m[4] * d.y; A.m[5] = R->m[5] * d.z;
  A.m[6] = R->m[6] * d.x; A.m[7] = R->m[7] * d.y; A.m[8] = R->m[8] * d.z;
  m3 I;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      I.m[3 * i + j] = (A.m[3 * i] * R->m[3 * j] + A.m[3 * i + 1] * R->m[3 * j + 1]) +
                       A.m[3 * i + 2] * R->m[3 * j + 2];
  return I;
}
/* Bullet btPlaneSpace1 restated: two unit tangents for a unit normal */
static inline void plane_space(v3 n, v3* p, v3* q) {
  if (fabsf(n.z) > 0.70710678f) {
    float a = n.y * n.y + n.z * n.z;
    float k = 1.0f / sqrtf(a);
    *p = V(0.0f, -n.z * k, n.y * k);
    *q = V(a * k, -n.x * p->z, n.x * p->y);
  } else {
    float a = n.x * n.x + n.y * n.y;
    float k = 1.0f / sqrtf(a);
    *p = V(-n.y * k, n.x * k, 0.0f);
    *q = V(-n.z * p->y, n.z * p->x, a * k);
  }
}
/* acos on [-1,1]: sqrt(1-|x|) * P7(|x|)  (Abramowitz & Stegun 4.4.46, |err| <= 2e-8).
 * A private polynomial (not libm) so that CPU and GPU agree bit for bit. */
float srlo_acosf(float x) {
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

/* ------------------------------------------------------------------ RNG */
static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
uint32_t srlo_rng(uint32_t key, uint32_t episode, uint32_t stream, uint32_t draw) {
  uint32_t h = mix32(key + 0x9E3779B9U);
  h = mix32(h ^ (episode + 0x85EBCA6BU));
  h = mix32(h ^ (stream + 0xC2B2AE35U));
  h = mix32(h ^ (draw + 0x27D4EB2FU));
  return h;
}
static inline uint32_t rng_below(uint32_t r, uint32_t n) {
  return (uint32_t)(((uint64_t)r * (uint64_t)n) >> 32);
}
uint32_t srlo_rng_below(uint32_t r, uint32_t n) { return rng_below(r, n); }
enum { STREAM_MESH = 0, STREAM_GOAL = 1, STREAM_ACTION = 2 };

/* ------------------------------------------------------------------ data */
typedef struct {
  int nv, nt;
  v3 v[MAXV];           /* vertices, COM frame (link vertex - com) */
  uint8_t tri[MAXT][3];
 float pl[MAXT][4];    /* face planes, COM frame: unit normal xyz, offset d (n.p <= d inside) */
  int ne;
  uint8_t edge[3 * MAXT / 2][4];   /* closed triangulated surface: edge = (vertex a < vertex b, the two faces sharing it) */
  v3 com;               /* URDF inertial origin (link frame) */
  float inv_mass;
  v3 inv_inertia;       /* body-frame diagonal: box inertia of the AABB (Bullet's default for
                           hull shapes when URDF inertia is not requested) */
  float radius;         /* max |v| (COM frame) */
} mesh_t;

typedef struct {
  v3 la, lb;            /* contact point in A / B body frames */
  v3 n;                 /* world normal, from B towards A */
  float dist;           /* signed distance (negative = penetration), margins removed */
  float in, it1, it2;   /* accumulated impulses (warm start) */
} mpoint_t;

typedef struct {
  int np;
  mpoint_t p[4];
  v3 axis;              /* cached GJK separating axis */
  int sn;               /* cached GJK simplex: number of vertex pairs, then (ia, ib) per pair */
  int sia[4], sib[4];
} manifold_t;

typedef struct {
  int np;
  int vid[8];
  float dist[8];
  float in[8], it1[8], it2[8];
} gmanifold_t;

typedef struct {
  /* rigid bodies (placed) */
  int nb;
  int mesh[MAXB];
  v3 x[MAXB]; q4 q[MAXB]; v3 v[MAXB]; v3 w[MAXB];
  v3 place_x[MAXB]; q4 place_q[MAXB];
  /* per-substep derived */
  m3 R[MAXB]; m3 Iw[MAXB];
  v3 wv[MAXB][MAXV];
  v3 amin[MAXB], amax[MAXB];
  /* contacts */
  gmanifold_t gm[MAXB];
  manifold_t man[MAXSLOT];
  int16_t slot_of_pair[NPAIR];
  int16_t pair_of_slot[MAXSLOT];    /* -1 = free */
  uint8_t slot_a[MAXSLOT], slot_b[MAXSLOT], colour[MAXSLOT];
  int ncolour;
  /* episode */
  int done;
  uint32_t episode;
  int ids[MAXB]; int list_pos;
  int pending;                      /* mesh id waiting at the spawn pose, -1 = none */
  int goal[4];                      /* u, v, h, w */
  float prev_metric[4];             /* Rewarder._memory */
  int substeps[2];
  int sweeps;                       /* solver sweeps of the last step (telemetry) */
  int status;
  int has_script; int script_ids[MAXB]; int script_goal[4];
  float* H;                         /* [res*res] */
  float* O;                         /* [ores*ores] */
} env_t;

struct srlo_env {
  srl_config c;
  int n_mesh;
  mesh_t* mesh;
  env_t* env;
  uint32_t seed;
  uint32_t sample_counter;
  /* derived constants */
  float px, inv_px;
  float lin_damp, ang_damp;
  int max_substeps;
  int goal_size, goal_min_h, goal_max_h, goal_min_w, goal_max_w;
  float goal_z;
  float scale;
  int AW, A;
  /* observable orientations of the pending rock (TestStackEnv, observer.py:127-140) */
  int n_orient;
  int n_slots;                      /* object maps per observation: n_orient, or episode_length * n_orient with ordering freedom */
  q4 orient_q[SRL_MAX_ORIENT];
};

/* ------------------------------------------------------------------ create */
/* `int(MAX_STEP_TIME / time_step)` (simulator.py:6, :46) */
static int c_max_substeps(const srl_config* c) {
  /* the quotient of the float32 time step nudged by 1e-6 of itself: 300 / 0.0125f = 23999.9996, the reference has 24000 */
  return c->max_substeps > 0 ? c->max_substeps : (int)(300.0 / (double)c->sim_time_step * (1.0 + 1e-6));
}

static int derive(struct srlo_env* e) {
  srl_config* c = &e->c;
  if (c->n_envs < 1 || c->episode_length < 1 || c->episode_length > MAXB)
    return fail(SRL_EINVAL, "n_envs/episode_length out of range");
  if (c->overhead_res < c->object_res || c->object_res < 2 || c->overhead_res > 256)
    return fail(SRL_EINVAL, "bad resolutions");
  e->px = c->object_max_dimension / (float)c->object_res;       /* env.py:136 */
  e->inv_px = (float)c->object_res / c->object_max_dimension;
  e->lin_damp = (float)pow(1.0 - (double)c->linear_damping, (double)c->sim_time_step);
  e->ang_damp = (float)pow(1.0 - (double)c->angular_damping, (double)c->sim_time_step);
  e->max_substeps = c_max_substeps(c);
  int H = c->overhead_res, h = c->object_res;
  /* rewarder.py:65-80 */
  e->goal_min_h = h; e->goal_min_w = h; e->goal_max_h = H; e->goal_max_w = H;
  e->goal_size = (int)((double)c->goal_size_ratio * H * H);
  if (e->goal_size <= 0) return fail(SRL_EINVAL, "goal_size_ratio must be a scalar in (0,1]");
  if (e->goal_size / e->goal_max_w > e->goal_min_h) e->goal_min_h = e->goal_size / e->goal_max_w;
  if (e->goal_size / e->goal_min_w < e->goal_max_h) e->goal_max_h = e->goal_size / e->goal_min_w;
  e->goal_z = c->max_z - c->object_max_dimension;               /* observer.py:378-382 */
  e->scale = c->reward_scale > 0.0f ? c->reward_scale : (float)c->episode_length; /* rewarder.py:97 */
  e->AW = H - h + 1;                                            /* env.py:207-211 */
  e->A = e->AW * e->AW;
  if (c->orientation_freedom < 0 || (1 << c->orientation_freedom) > SRL_MAX_ORIENT)
    return fail(SRL_EINVAL, "orientation_freedom must be in 0..4");
  e->n_orient = 1 << c->orientation_freedom;                    /* observer.py:127 */
  if (c->ordering_freedom != 0 && c->ordering_freedom != 1) return fail(SRL_EINVAL, "ordering_freedom must be 0 or 1");
  e->n_slots = c->ordering_freedom ? c->episode_length * e->n_orient : e->n_orient;
  for (int i = 0; i < e->n_orient; ++i) {   /* inverse of getQuaternionFromEuler([0, 0, i 2 pi / n]), observer.py:129-139 */
    double half = -0.5 * ((double)i * 2.0 * 3.14159265358979323846 / (double)e->n_orient);
    e->orient_q[i].x = 0.0f; e->orient_q[i].y = 0.0f;
    e->orient_q[i].z = i == 0 ? 0.0f : (float)sin(half); e->orient_q[i].w = i == 0 ? 1.0f : (float)cos(half);
  }
  return SRL_OK;
}

int srlo_create(const srl_config* cfg, srlo_env** out) {
  struct srlo_env* e = (struct srlo_env*)calloc(1, sizeof *e);
  if (!e) return fail(SRL_EINVAL, "oom");
  e->c = *cfg;
  int rc = derive(e);
  if (rc) { free(e); return rc; }
  e->env = (env_t*)calloc((size_t)cfg->n_envs, sizeof(env_t));
  if (!e->env) { free(e); return fail(SRL_EINVAL, "oom"); }
  for (int i = 0; i < cfg->n_envs; ++i) {
    env_t* s = &e->env[i];
    s->H = (float*)calloc((size_t)cfg->overhead_res * cfg->overhead_res, sizeof(float));
    s->O = (float*)calloc((size_t)cfg->object_res * cfg->object_res * SRL_MAX_ORIENT * (cfg->ordering_freedom ? SRL_MAX_BODIES : 1), sizeof(float));
    s->done = 1;                                                /* env.py:219-220 */
    s->pending = -1;
    for (int k = 0; k < NPAIR; ++k) s->slot_of_pair[k] = -1;
    for (int k = 0; k < MAXSLOT; ++k) s->pair_of_slot[k] = -1;
  }
  *out = e;
  return SRL_OK;
}

void srlo_destroy(srlo_env* e) {
  if (!e) return;
  for (int i = 0; i < e->c.n_envs; ++i) { free(e->env[i].H); free(e->env[i].O); }
  free(e->env); free(e->mesh); free(e);
}

int srlo_load_meshes(srlo_env* e, const float* verts, const int32_t* vert_off,
                     const int32_t* tris, const int32_t* tri_off, const float* mass_com,
                     int32_t n_mesh) {
  if (n_mesh < 1) return fail(SRL_EINVAL, "empty mesh pool");     /* env.py:103 */
  free(e->mesh);
  e->mesh = (mesh_t*)calloc((size_t)n_mesh, sizeof(mesh_t));
  e->n_mesh = n_mesh;
  for (int m = 0; m < n_mesh; ++m) {
    mesh_t* M = &e->mesh[m];
    M->nv = vert_off[m + 1] - vert_off[m];
    M->nt = tri_off[m + 1] - tri_off[m];
    if (M->nv < 4 || M->nv > MAXV || M->nt < 4 || M->nt > MAXT)
      return fail(SRL_EINVAL, "mesh exceeds SRL_MAX_VERTS/SRL_MAX_TRIS");
    float mass = mass_com[4 * m];
    M->com = V(mass_com[4 * m + 1], mass_com[4 * m + 2], mass_com[4 * m + 3]);
    v3 lo = V(1e30f, 1e30f, 1e30f), hi = V(-1e30f, -1e30f, -1e30f);
    float r2 = 0.0f;
    for (int k = 0; k < M->nv; ++k) {
      const float* p = verts + 3 * (size_t)(vert_off[m] + k);
      v3 a = vsub(V(p[0], p[1], p[2]), M->com);
      M->v[k] = a;
      lo = V(fminf(lo.x, a.x), fminf(lo.y, a.y), fminf(lo.z, a.z));
      hi = V(fmaxf(hi.x, a.x), fmaxf(hi.y, a.y), fmaxf(hi.z, a.z));
      float d2 = vdot(a, a);
      if (d2 > r2) r2 = d2;
    }
    M->radius = sqrtf(r2);
    for (int k = 0; k < M->nt; ++k) {
      const int32_t* t = tris + 3 * (size_t)(tri_off[m] + k);
      for (int j = 0; j < 3; ++j) {
        if (t[j] < 0 || t[j] >= M->nv) return fail(SRL_EINVAL, "triangle index out of range");
        M->tri[k][j] = (uint8_t)t[j];
      }
      v3 a = M->v[t[0]], b = M->v[t[1]], c = M->v[t[2]];
      v3 n = vcross(vsub(b, a), vsub(c, a));
      float len = sqrtf(vdot(n, n));
      if (!(len > 0.0f)) return fail(SRL_EINVAL, "degenerate triangle");
      n = vscale(n, 1.0f / len);
      M->pl[k][0] = n.x; M->pl[k][1] = n.y; M->pl[k][2] = n.z; M->pl[k][3] = vdot(n, a);
    }
    {   /* edge list (the overhead renderer's silhouette test): every edge of the closed surface has exactly two faces */
      static int first[MAXV * MAXV];
      for (int q = 0; q < M->nv * M->nv; ++q) first[q] = -1;
      M->ne = 0;
      for (int k = 0; k < M->nt; ++k)
        for (int j = 0; j < 3; ++j) {
          int a = M->tri[k][j], b = M->tri[k][(j + 1) % 3];
          if (a > b) { int tmp = a; a = b; b = tmp; }
          if (a == b) return fail(SRL_EINVAL, "degenerate triangle");
          int* slot = &first[a * M->nv + b];
          if (*slot == -1) *slot = k;
          else if (*slot >= 0) {
            uint8_t* e4 = M->edge[M->ne++];
            e4[0] = (uint8_t)a; e4[1] = (uint8_t)b; e4[2] = (uint8_t)*slot; e4[3] = (uint8_t)k;
            *slot = -2;
          } else return fail(SRL_EINVAL, "mesh is not a closed two-manifold (an edge has more than two faces)");
        }
      for (int q = 0; q < M->nv * M->nv; ++q)
        if (first[q] >= 0) return fail(SRL_EINVAL, "mesh is not a closed surface (an edge has only one face)");
    }
    /* Bullet btCompoundShape/btPolyhedralConvexShape::calculateLocalInertia restated:
     * inertia of the solid box spanned by the shape's AABB (pybullet ignores the URDF
     * inertia unless URDF_USE_INERTIA_FROM_FILE is passed; simulator.py:300 passes no flags). */
    float lx = hi.x - lo.x, ly = hi.y - lo.y, lz = hi.z - lo.z;
    float k12 = mass / 12.0f;
    v3 I = V(k12 * (ly * ly + lz * lz), k12 * (lx * lx + lz * lz), k12 * (lx * lx + ly * ly));
    M->inv_mass = 1.0f / mass;
    M->inv_inertia = V(1.0f / I.x, 1.0f / I.y, 1.0f / I.z);
  }
  return SRL_OK;
}

int srlo_seed(srlo_env* e, uint32_t seed) {
  e->seed = seed;
  e->sample_counter = 0;
  for (int i = 0; i < e->c.n_envs; ++i) e->env[i].episode = 0;
  return SRL_OK;
}

int srlo_set_script(srlo_env* e, const int32_t* mesh_ids, const int32_t* goal_rect) {
  int L = e->c.episode_length;
  for (int i = 0; i < e->c.n_envs; ++i) {
    env_t* s = &e->env[i];
    for (int k = 0; k < L; ++k) {
      int id = mesh_ids[(size_t)i * L + k];
      if (id < 0 || id >= e->n_mesh) return fail(SRL_EINVAL, "script mesh id out of range");
      s->script_ids[k] = id;
    }
    for (int k = 0; k < 4; ++k) s->script_goal[k] = goal_rect[4 * i + k];
    s->has_script = 1;
  }
  return SRL_OK;
}

/* ------------------------------------------------------------------ goal (rewarder.py:211-259) */
/* `Rewarder._reset_goal`, scalar-area branch (rewarder.py:225-259), on an explicit draw list — what the reference's RNG
 * stream is asked for, in its order: x24 = the value of `beta(b, 4 - b)` as a 24-bit fraction (the swap bit `randint(2)`
 * only selects which Beta it is drawn from), ru / rv = 32-bit words behind `randint(lo, hi)` of the two offsets (reduced
 * to the range by rng_below).  Pinned by the reference's own rewarder.py (tests/golden/rewarder_golden.npz). */
void srlo_goal_from_draws(const srl_config* cfg, uint32_t x24, uint32_t ru, uint32_t rv, int32_t* rect) {
  struct srlo_env tmp;
  memset(&tmp, 0, sizeof tmp);
  tmp.c = *cfg;
  derive(&tmp);
  int H = cfg->overhead_res;
  int h = tmp.goal_min_h + (int)(((uint64_t)x24 * (uint64_t)(tmp.goal_max_h - tmp.goal_min_h)) >> 24);   /* int(min + beta (max - min)) */
  int w = tmp.goal_size / h;
  if (w < tmp.goal_min_w) w = tmp.goal_min_w;
  if (w > tmp.goal_max_w) w = tmp.goal_max_w;
  int umax = H - h, vmax = H - w;
  int ulo = umax / 8, uhi = 7 * umax / 8 + 1;   /* margin_factor = 8, rewarder.py:16 */
  int vlo = vmax / 8, vhi = 7 * vmax / 8 + 1;
  int u = ulo + (int)rng_below(ru, (uint32_t)(uhi - ulo));
  int v = vlo + (int)rng_below(rv, (uint32_t)(vhi - vlo));
  rect[0] = u; rect[1] = v; rect[2] = h; rect[3] = w;
}

void srlo_goal_from_rng(const srl_config* cfg, uint32_t key, uint32_t episode, int32_t* rect) {
  /* b = 1 + 2*randint(2); beta(b, 4-b): Beta(1,3) = min of 3 uniforms, Beta(3,1) = max of 3 */
  uint32_t bbit = srlo_rng(key, episode, STREAM_GOAL, 0) >> 31;
  uint32_t u0 = srlo_rng(key, episode, STREAM_GOAL, 1) >> 8;
  uint32_t u1 = srlo_rng(key, episode, STREAM_GOAL, 2) >> 8;
  uint32_t u2 = srlo_rng(key, episode, STREAM_GOAL, 3) >> 8;
  uint32_t lo = u0 < u1 ? u0 : u1; lo = lo < u2 ? lo : u2;
  uint32_t hi = u0 > u1 ? u0 : u1; hi = hi > u2 ? hi : u2;
  uint32_t X = bbit ? hi : lo; /* 24-bit fixed point in [0,1) */
  srlo_goal_from_draws(cfg, X, srlo_rng(key, episode, STREAM_GOAL, 4), srlo_rng(key, episode, STREAM_GOAL, 5), rect);
}

/* ------------------------------------------------------------------ depth codec */
/* pybullet TinyRenderer depth (restated): d = far (t - near) / (t (far - near)), t = eye distance */
static inline float depth_encode(float t, float nearp, float farp) {
  if (t < nearp) t = nearp;
  if (t > farp) t = farp;
  return (farp * (t - nearp)) / (t * (farp - nearp));
}
/* observer.py:259-260 with numpy's float32 arithmetic (python scalars are weak) */
static inline float elev_overhead(const srl_config* c, float d) {
  float num = (float)((double)FAR_PLANE * ((double)FAR_PLANE - (double)c->max_z));
  return FAR_PLANE - num / (FAR_PLANE - c->max_z * d);
}
/* observer.py:274-275 */
static inline float elev_object(const srl_config* c, float d) {
  double oz = (double)c->object_max_dimension; /* _object_z = max(_object_x,_object_y), observer.py:79-81 */
  float c1 = (float)((double)FAR_PLANE + oz / 2);
  float c2 = (float)((double)FAR_PLANE * (double)FAR_PLANE - (oz / 2) * (oz / 2));
  return c1 - c2 / (FAR_PLANE + c->object_max_dimension * (0.5f - d));
}
void srlo_depth_to_elevation(const srl_config* c, int which, const float* depth, float* elev) {
  if (which == 0) {
    int n = c->overhead_res * c->overhead_res;
    for (int i = 0; i < n; ++i) elev[i] = elev_overhead(c, depth[i]);
  } else {
    int r = c->object_res;
    for (int i = 0; i < r; ++i)
      for (int j = 0; j < r; ++j)                       /* d[:, ::-1], observer.py:277 */
        elev[i * r + j] = elev_object(c, depth[i * r + (r - 1 - j)]);
  }
}

/* ------------------------------------------------------------------ renderer (convex ray cast) */
/* A rock is the intersection of its face half-spaces.  Along the vertical line through a pixel centre
 * the hull spans [z_lo, z_hi]: z_hi = min over up-facing planes (n_z >= 0), z_lo = max over down-facing
 * planes (n_z < 0).  |n_z| is clamped to >= 1e-6, so a vertical face acts as an up/down plane of enormous
 * slope: it never limits z on its inner side and empties the interval on its outer side.
 * The object camera (from below) sees z_lo where z_lo <= z_hi.
 * The overhead camera sees z_hi where the pixel centre lies inside the rock's silhouette: the convex polygon whose
 * sides are the projections of the edges shared by an up-facing and a down-facing face.  Side through the projected
 * end points A, B (A the lower vertex index): E(p) = fma(ea, p.x, fma(eb, p.y, ec)), ea = A.y - B.y, eb = B.x - A.x,
 * ec = -fma(ea, A.x, eb * A.y), all three negated if E(centre of mass) < 0; inside iff E(p) >= 0 for every side.
 * (Same set as z_lo <= z_hi up to pixel centres on the outline; a third of the per-pixel work.) */
typedef struct { float a, b, c; int type; } rplane_t;   /* z = a x + b y + c; type 0 up, 1 down */

static int make_rplanes(const mesh_t* M, const m3* R, v3 x, rplane_t* out) {
  for (int t = 0; t < M->nt; ++t) {
    v3 nw = mmul(R, V(M->pl[t][0], M->pl[t][1], M->pl[t][2]));
    float dw = M->pl[t][3] + vdot(nw, x);
    rplane_t p;
    float nz = nw.z;
    if (nz >= 0.0f) { if (nz < 1e-6f) nz = 1e-6f; p.type = 0; }
    else { if (nz > -1e-6f) nz = -1e-6f; p.type = 1; }
    p.a = -nw.x / nz; p.b = -nw.y / nz; p.c = dw / nz;
    out[t] = p;
  }
  return M->nt;
}

/* returns 1 and [zlo, zhi] if the vertical line through (px, py) meets the hull */
static inline int ray_cast(const rplane_t* pl, int n, float px, float py, float* zlo, float* zhi) {
  float hi = 1e30f, lo = -1e30f;
  for (int t = 0; t < n; ++t) {   /* fused multiply-adds: part of the definition (the kernel uses v_fma_f32) */
    float z = fmaf(pl[t].a, px, fmaf(pl[t].b, py, pl[t].c));
    if (pl[t].type == 0) hi = fminf(hi, z);
    else lo = fmaxf(lo, z);
  }
  *zlo = lo; *zhi = hi;
  return lo <= hi;
}

/* pixel range [i0, i1] whose centres lie in [lo, hi] (clipped to the map) */
static inline int pixel_range(float lo, float hi, float inv_px, int res, int* i0, int* i1) {
  float f0 = ceilf(lo * inv_px - 0.5f), f1 = floorf(hi * inv_px - 0.5f);
  if (f0 < 0.0f) f0 = 0.0f;
  if (f1 > (float)(res - 1)) f1 = (float)(res - 1);
  if (f1 < f0) return 0;
  *i0 = (int)f0; *i1 = (int)f1;
  return 1;
}

/* Which faces and outline sides a pixel consults (round 4; the kernels' ray cast visits a rock in items of ITEM_ROWS pixel
 * rows x 2 columns, csrc/stage.h stage_compute).  Along the vertical line through a pixel inside the outline the hull's top
 * is the face the line pierces, so the minimum over ALL up-facing planes equals the minimum over any subset that holds that
 * face — up to the last bit where two faces are coplanar within rounding, which is why the subset is part of the definition
 * and not an optimisation behind it:
 *   item row r of a rock = pixel rows i0 + ITEM_ROWS r .. (i0 the first row of its bounding box); with more than ROW_SPANS
 *   item rows every pixel consults every face and side (as before round 4);
 *   a face reaches the item rows [r0, r1] of the pixel rows from the last one whose centre is at or below its x extent to the
 *   first one at or above it (slab_rows); so does an outline side with the x extent of its edge;
 *   a pixel of item row r consults the faces whose first row r0 lies in [R(r), r], R(r) = the smallest r0 among the faces
 *   that reach r (r1 >= r) — a superset of those that reach r, and one contiguous range of the faces ordered by r0, whatever
 *   the order inside one r0 (the kernel fills its lists with LDS counters); the sides likewise;
 *   and within item row r only the columns between the bounds the outline can reach there are visited at all (span_columns:
 *   over the rows' x range [xa, xb] a side with eb > 0 bounds y from below by the smaller of its line's values at xa, xb,
 *   one with eb < 0 from above by the larger; widened by 1e-4 m and to the column pairs the items cover). */
enum { ITEM_ROWS = 4, ROW_SPANS = 16 };

static void slab_rows(float x0, float x1, float inv_px, int i0, int nirows, int* r0, int* r1) {
  float f0 = floorf(x0 * inv_px - 0.5f) - (float)i0, f1 = ceilf(x1 * inv_px - 0.5f) - (float)i0;
  int a = f0 > 0.0f ? (int)(f0 * (1.0f / ITEM_ROWS)) : 0, z = f1 > 0.0f ? (int)(f1 * (1.0f / ITEM_ROWS)) : 0;
  if (a > nirows - 1) a = nirows - 1;
  if (z > nirows - 1) z = nirows - 1;
  *r0 = a; *r1 = z;
}

/* R[r] = the smallest first row among the entries that reach item row r (last row >= r); ROW_SPANS + 1 if none does */
static void first_rows(int n, const int* r0, const int* r1, int* R) {
  for (int r = 0; r < ROW_SPANS; ++r) R[r] = ROW_SPANS + 1;
  for (int k = 0; k < n; ++k)
    for (int r = 0; r <= r1[k]; ++r) if (r0[k] < R[r]) R[r] = r0[k];
}

/* columns [*ja, *jz] of the bounding box [j0, j1] visited in the item row whose pixel rows are ia .. ib; returns 0 if none */
static int span_columns(const struct srlo_env* e, int ns, const float* ea, const float* eb, const float* ec, int ia, int ib,
                        int j0, int j1, int* ja, int* jz) {
  float xa = ((float)ia + 0.5f) * e->px, xe = ((float)ib + 0.5f) * e->px;
  float ylo = -1e30f, yhi = 1e30f;
  int empty = 0;
  for (int k = 0; k < ns; ++k) {
    float fa = fmaf(ea[k], xa, ec[k]), fe = fmaf(ea[k], xe, ec[k]);
    if (fabsf(eb[k]) < 1e-12f) { empty = empty || (fa < 0.0f && fe < 0.0f); continue; }
    float rcp = -1.0f / eb[k];
    float la = fa * rcp, le = fe * rcp;
    if (eb[k] > 0.0f) ylo = fmaxf(ylo, fminf(la, le)); else yhi = fminf(yhi, fmaxf(la, le));
  }
  float fl = ceilf((ylo - 1e-4f) * e->inv_px - 0.5f), fh = floorf((yhi + 1e-4f) * e->inv_px - 0.5f);
  int a = fl > (float)j0 ? (fl < (float)(j1 + 1) ? (int)fl : j1 + 1) : j0;
  int z = fh < (float)j1 ? (fh > (float)(j0 - 1) ? (int)fh : j0 - 1) : j1;
  if (empty || a > z) return 0;
  int cnt = (z - a + 2) >> 1;                         /* items of two columns from column a */
  *ja = a; *jz = a + 2 * cnt - 1 < j1 ? a + 2 * cnt - 1 : j1;
  return 1;
}

/* O1: overhead height map of the placed bodies (observer.py:252-260; row <-> +x, col <-> +y) */
static void render_heightmap_raw(const struct srlo_env* e, int nb, const int* mesh, const v3* x,
                                 const q4* q, float* H) {
  const srl_config* c = &e->c;
  int res = c->overhead_res;
  for (int k = 0; k < res * res; ++k) H[k] = 0.0f;
  for (int b = 0; b < nb; ++b) {
    const mesh_t* M = &e->mesh[mesh[b]];
    m3 R = quat_to_mat(q[b]);
    float wx[MAXV], wy[MAXV];
    float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
    for (int k = 0; k < M->nv; ++k) {
      v3 a = mmul_add(&R, M->v[k], x[b]);
      wx[k] = a.x; wy[k] = a.y;
      xmin = fminf(xmin, a.x); xmax = fmaxf(xmax, a.x);
      ymin = fminf(ymin, a.y); ymax = fmaxf(ymax, a.y);
    }
    int i0, i1, j0, j1;
    if (!pixel_range(xmin, xmax, e->inv_px, res, &i0, &i1)) continue;
    if (!pixel_range(ymin, ymax, e->inv_px, res, &j0, &j1)) continue;
    const int nirows = (i1 - i0 + ITEM_ROWS) / ITEM_ROWS;
    const int nir = nirows <= ROW_SPANS ? nirows : 0;
    rplane_t pl[MAXT];
    int np = make_rplanes(M, &R, x[b], pl);
    /* up-facing faces in face order, with the item rows they reach */
    rplane_t up[MAXT];
    int ur0[MAXT], ur1[MAXT], nup = 0;
    for (int t = 0; t < np; ++t) {
      if (pl[t].type != 0) continue;
      const uint8_t* tv = M->tri[t];
      float fa = wx[tv[0]], fb = wx[tv[1]], fc = wx[tv[2]];
      ur0[nup] = 0; ur1[nup] = 0;
      if (nir) slab_rows(fminf(fa, fminf(fb, fc)), fmaxf(fa, fmaxf(fb, fc)), e->inv_px, i0, nirows, &ur0[nup], &ur1[nup]);
      up[nup++] = pl[t];
    }
    /* outline sides in edge order (at most nt + 2 - nup of them: a closed triangulated cap with an s-edge rim has >= s - 2
     * triangles) */
    float ea[MAXT + 2], eb[MAXT + 2], ec[MAXT + 2];
    int sr0[MAXT + 2], sr1[MAXT + 2], ns = 0;
    const int cap = M->nt + 2 - nup;
    for (int k = 0; k < M->ne; ++k) {
      const uint8_t* e4 = M->edge[k];
      if (pl[e4[2]].type == pl[e4[3]].type) continue;       /* not on the outline */
      if (ns >= cap) continue;
      float Ax = wx[e4[0]], Ay = wy[e4[0]], Bx = wx[e4[1]], By = wy[e4[1]];
      float a = Ay - By, bb = Bx - Ax;
      float cc = -fmaf(a, Ax, bb * Ay);
      float s = fmaf(a, x[b].x, fmaf(bb, x[b].y, cc));      /* the centre of mass is inside */
      if (s < 0.0f) { a = -a; bb = -bb; cc = -cc; }
      sr0[ns] = 0; sr1[ns] = 0;
      if (nir) slab_rows(fminf(Ax, Bx), fmaxf(Ax, Bx), e->inv_px, i0, nirows, &sr0[ns], &sr1[ns]);
      ea[ns] = a; eb[ns] = bb; ec[ns] = cc; ++ns;
    }
    int Rp[ROW_SPANS], Rs[ROW_SPANS];
    first_rows(nup, ur0, ur1, Rp);
    first_rows(ns, sr0, sr1, Rs);
    for (int r = 0; r < nirows; ++r) {
      const int ia = i0 + ITEM_ROWS * r, ib = ia + ITEM_ROWS - 1 < i1 ? ia + ITEM_ROWS - 1 : i1;
      int ja = j0, jz = j1;
      const int pr0 = nir ? Rp[r] : 0, sr0_ = nir ? Rs[r] : 0, rr = nir ? r : 0;   /* consulted: first row in [.., rr] */
      if (nir && !span_columns(e, ns, ea, eb, ec, ia, ib, j0, j1, &ja, &jz)) continue;
      if (pr0 > rr || sr0_ > rr || nup == 0 || ns == 0) continue;   /* a row the lists do not reach holds no rock pixel */
      for (int i = ia; i <= ib; ++i) {
        float px = ((float)i + 0.5f) * e->px;
        for (int j = ja; j <= jz; ++j) {
          float py = ((float)j + 0.5f) * e->px;
          float hi = 1e30f, lo = 1e30f;
          for (int t = 0; t < nup; ++t)
            if (ur0[t] >= pr0 && ur0[t] <= rr) hi = fminf(hi, fmaf(up[t].a, px, fmaf(up[t].b, py, up[t].c)));
          for (int t = 0; t < ns; ++t)
            if (sr0[t] >= sr0_ && sr0[t] <= rr) lo = fminf(lo, fmaf(ea[t], px, fmaf(eb[t], py, ec[t])));
          if (lo >= 0.0f && hi > H[i * res + j]) H[i * res + j] = hi;
        }
      }
    }
  }
}

/* heights above the ground -> what Observer.state[0] holds: TinyRenderer's depth encoding and the reference's decode */
static void heightmap_codec(const struct srlo_env* e, float* H) {
  const srl_config* c = &e->c;
  const int res = c->overhead_res;
  float nearp = FAR_PLANE - c->max_z;
  for (int k = 0; k < res * res; ++k) {
    float d = depth_encode(FAR_PLANE - H[k], nearp, FAR_PLANE);
    H[k] = elev_overhead(c, d);
  }
}

static void render_heightmap(const struct srlo_env* e, int nb, const int* mesh, const v3* x, const q4* q, float* H) {
  render_heightmap_raw(e, nb, mesh, x, q, H);
  heightmap_codec(e, H);
}

/* O1, THE PLAIN STATEMENT (observer.py:252-260: one getCameraImage call of the overhead camera).  Kept apart from
 * render_heightmap on purpose and never to be touched by a render optimisation: no item rows, no list ranges, no column
 * spans — every pixel of a rock's bounding box consults every face / every outline side.  tests/test_render_plain.py holds
 * render_heightmap (the culled definition above, which the kernels restate) to this function on random scenes, and the
 * `-m gpu` tests hold srl_render_heightmap to it.
 *   form 1: max over rocks of (min over ALL up-facing planes) where every outline side's E(p) >= 0   (round-3 definition)
 *   form 2: max over rocks of z_hi where z_lo <= z_hi over ALL planes (no edge table at all)          (round-1 definition)
 * raw != 0: heights before the depth codec (where the culled definition may differ in the last bit on coplanar faces);
 * raw == 0: after the codec, as Observer.state[0] holds them. */
static void render_heightmap_all(const struct srlo_env* e, int nb, const int* mesh, const v3* x, const q4* q, int form,
                                 int raw, float* H) {
  const srl_config* c = &e->c;
  const int res = c->overhead_res;
  for (int k = 0; k < res * res; ++k) H[k] = 0.0f;
  for (int b = 0; b < nb; ++b) {
    const mesh_t* M = &e->mesh[mesh[b]];
    m3 R = quat_to_mat(q[b]);
    float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
    for (int k = 0; k < M->nv; ++k) {
      v3 a = mmul_add(&R, M->v[k], x[b]);
      xmin = fminf(xmin, a.x); xmax = fmaxf(xmax, a.x);
      ymin = fminf(ymin, a.y); ymax = fmaxf(ymax, a.y);
    }
    int i0, i1, j0, j1;
    if (!pixel_range(xmin, xmax, e->inv_px, res, &i0, &i1)) continue;
    if (!pixel_range(ymin, ymax, e->inv_px, res, &j0, &j1)) continue;
    rplane_t pl[MAXT];
    const int np = make_rplanes(M, &R, x[b], pl);
    float ea[3 * MAXT / 2 + 3], eb[3 * MAXT / 2 + 3], ec[3 * MAXT / 2 + 3];
    int ns = 0;
    if (form == 1)
      for (int k = 0; k < M->ne; ++k) {
        const uint8_t* e4 = M->edge[k];
        if (pl[e4[2]].type == pl[e4[3]].type) continue;       /* not on the outline */
        v3 A = mmul_add(&R, M->v[e4[0]], x[b]), B = mmul_add(&R, M->v[e4[1]], x[b]);
        float a = A.y - B.y, bb = B.x - A.x;
        float cc = -fmaf(a, A.x, bb * A.y);
        float s = fmaf(a, x[b].x, fmaf(bb, x[b].y, cc));      /* the centre of mass is inside */
        if (s < 0.0f) { a = -a; bb = -bb; cc = -cc; }
        ea[ns] = a; eb[ns] = bb; ec[ns] = cc; ++ns;
      }
    for (int i = i0; i <= i1; ++i) {
      float px = ((float)i + 0.5f) * e->px;
      for (int j = j0; j <= j1; ++j) {
        float py = ((float)j + 0.5f) * e->px;
        if (form == 1) {
          float hi = 1e30f, lo = 1e30f;
          for (int t = 0; t < np; ++t)
            if (pl[t].type == 0) hi = fminf(hi, fmaf(pl[t].a, px, fmaf(pl[t].b, py, pl[t].c)));
          for (int t = 0; t < ns; ++t) lo = fminf(lo, fmaf(ea[t], px, fmaf(eb[t], py, ec[t])));
          if (ns > 0 && lo >= 0.0f && hi > H[i * res + j]) H[i * res + j] = hi;
        } else {
          float lo, hi;
          if (ray_cast(pl, np, px, py, &lo, &hi) && hi > H[i * res + j]) H[i * res + j] = hi;
        }
      }
    }
  }
  if (!raw) heightmap_codec(e, H);
}

/* O2: underside map of a mesh at the spawn pose (link frame at spawn, identity orientation;
 * observer.py:262-277).  O = (z_c + oz/2) - z_underside, 0 where empty. */
/* one observable orientation oi of the pending rock: it turns about its link-frame origin at the centre of the map */
static void render_object_1(const struct srlo_env* e, int mesh_id, int oi, float* O) {
  const srl_config* c = &e->c;
  int r = c->object_res;
  float half = c->object_max_dimension * 0.5f;
  for (int k = 0; k < r * r; ++k) O[k] = 1e30f;
  if (mesh_id >= 0) {
    const mesh_t* M = &e->mesh[mesh_id];
    m3 I; for (int k = 0; k < 9; ++k) I.m[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    /* map coordinates: link frame shifted so that the map starts at 0 */
    v3 xs = V(M->com.x + half, M->com.y + half, M->com.z);
    if (e->n_orient > 1) {
      I = quat_to_mat(e->orient_q[oi]);
      xs = vadd(mmul(&I, M->com), V(half, half, 0.0f));
    }
    rplane_t pl[MAXT];
    int np = make_rplanes(M, &I, xs, pl);
    for (int i = 0; i < r; ++i) {
      float px = ((float)i + 0.5f) * e->px;
      for (int j = 0; j < r; ++j) {
        float py = ((float)j + 0.5f) * e->px;
        float lo, hi;
        if (ray_cast(pl, np, px, py, &lo, &hi)) O[i * r + j] = lo;
      }
    }
  }
  float nearp = FAR_PLANE - half, farp = FAR_PLANE + half;
  for (int k = 0; k < r * r; ++k) {
    float d = O[k] > 1e29f ? 1.0f : depth_encode(FAR_PLANE + O[k], nearp, farp);
    O[k] = elev_object(c, d);
  }
}

static void render_object(const struct srlo_env* e, int mesh_id, float* O) {
  int rr = e->c.object_res * e->c.object_res;
  for (int oi = 0; oi < e->n_orient; ++oi) render_object_1(e, mesh_id, oi, O + (size_t)oi * rr);
}

/* ------------------------------------------------------------------ pose (observer.py:392-421) */
void srlo_pose(const srl_config* c, const float* H, const float* O, int32_t u, int32_t v, float* xyz) {
  int res = c->overhead_res, r = c->object_res;
  float px = c->object_max_dimension / (float)c->object_res;
  float z = -1e30f; /* np.max over an empty selection raises in the reference; never empty for a real rock */
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < r; ++j) {
      float o = O[i * r + j];
      if (o > 1e-4f) {
        float s = H[(u + i) * res + (v + j)] + o;
        if (s > z) z = s;
      }
    }
  float half = ((float)r * px) * 0.5f;   /* _object_x/2 = _object_z/2 */
  xyz[0] = (float)u * px + half;
  xyz[1] = (float)v * px + half;
  xyz[2] = z - half;
}

/* ------------------------------------------------------------------ reward sums (rewarder.py:297-307) */
/* Fixed order: pixels in groups of 4 (row-major), group g accumulates into partial[g % 512],
 * then a halving tree over the 512 partials. */
#define NPART 512
void srlo_iou_sums(const srl_config* c, const float* H, const int32_t* g, float* inter, float* uni) {
  int res = c->overhead_res;
  float gz = c->max_z - c->object_max_dimension;
  float pi[NPART], pu[NPART];
  for (int k = 0; k < NPART; ++k) { pi[k] = 0.0f; pu[k] = 0.0f; }
  int n = res * res;
  for (int base = 0; base < n; base += 4) {
    int t = (base >> 2) & (NPART - 1);
    for (int k = base; k < base + 4; ++k) {
      int i = k / res, j = k % res;
      int in = (i >= g[0] && i < g[0] + g[2] && j >= g[1] && j < g[1] + g[3]);
      float h = H[k];
      if (in) {
        pi[t] += fminf(h, gz);
        pu[t] += fmaxf(h, gz);
      } else {
        pu[t] += fmaxf(h, 0.0f);
      }
    }
  }
  for (int s = NPART / 2; s >= 1; s >>= 1)
    for (int t = 0; t < s; ++t) { pi[t] += pi[t + s]; pu[t] += pu[t + s]; }
  *inter = pi[0];
  *uni = pu[0];
}

/* ================================================================== physics */
/* ---- closest point of a simplex to the origin (Ericson, Real-Time Collision Detection 5.1) */
typedef struct {
  v3 w[4], p[4], q[4];
  int ia[4], ib[4];
  int n;
} simplex_t;

static void closest_tri(v3 a, v3 b, v3 c, float* l /*3*/, int* used) {
  v3 ab = vsub(b, a), ac = vsub(c, a), ap = vneg(a);
  float d1 = vdot(ab, ap), d2 = vdot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) { l[0] = 1.0f; l[1] = 0.0f; l[2] = 0.0f; *used = 1; return; }
  v3 bp = vneg(b);
  float d3 = vdot(ab, bp), d4 = vdot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) { l[0] = 0.0f; l[1] = 1.0f; l[2] = 0.0f; *used = 2; return; }
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
    float v = d1 / (d1 - d3);
    l[0] = 1.0f - v; l[1] = v; l[2] = 0.0f; *used = 3; return;
  }
  v3 cp = vneg(c);
  float d5 = vdot(ab, cp), d6 = vdot(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) { l[0] = 0.0f; l[1] = 0.0f; l[2] = 1.0f; *used = 4; return; }
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
    float w = d2 / (d2 - d6);
    l[0] = 1.0f - w; l[1] = 0.0f; l[2] = w; *used = 5; return;
  }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
    float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    l[0] = 0.0f; l[1] = 1.0f - w; l[2] = w; *used = 6; return;
  }
  float denom = 1.0f / ((va + vb) + vc);
  float v = vb * denom, w = vc * denom;
  l[0] = (1.0f - v) - w; l[1] = v; l[2] = w; *used = 7;
}

/* 1: origin is on the outer side of plane abc (w.r.t. d); 0: inner side; -1: degenerate */
static int outside_plane(v3 a, v3 b, v3 c, v3 d) {
  v3 n = vcross(vsub(b, a), vsub(c, a));
  float sp = vdot(vneg(a), n);
  float sd = vdot(vsub(d, a), n);
  if (sd * sd < 1e-24f) return -1;
  return (sp * sd < 0.0f) ? 1 : 0;
}

/* Reduce simplex to the feature closest to the origin; lam/ v out.
 * returns 1 ok, 0 degenerate, 2 origin enclosed (tetrahedron). */
static int simplex_closest(simplex_t* s, float* lam, v3* vout) {
  int used = 0;
  float l[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (s->n == 1) {
    l[0] = 1.0f; used = 1;
  } else if (s->n == 2) {
    v3 a = s->w[0], b = s->w[1];
    v3 ab = vsub(b, a);
    float t = vdot(vneg(a), ab);
    if (t <= 0.0f) { l[0] = 1.0f; used = 1; }
    else {
      float den = vdot(ab, ab);
      if (t >= den) { l[1] = 1.0f; used = 2; }
      else { t = t / den; l[0] = 1.0f - t; l[1] = t; used = 3; }
    }
  } else if (s->n == 3) {
    closest_tri(s->w[0], s->w[1], s->w[2], l, &used);
  } else {
    v3 a = s->w[0], b = s->w[1], c = s->w[2], d = s->w[3];
    int o0 = outside_plane(a, b, c, d);
    int o1 = outside_plane(a, c, d, b);
    int o2 = outside_plane(a, d, b, c);
    int o3 = outside_plane(b, d, c, a);
    if (o0 < 0 || o1 < 0 || o2 < 0 || o3 < 0) return 0;
    if (!o0 && !o1 && !o2 && !o3) return 2;
    float best = 1e30f;
    float tl[3]; int tu;
    if (o0) {
      closest_tri(a, b, c, tl, &tu);
      v3 p = vmadd(vmadd(vscale(a, tl[0]), b, tl[1]), c, tl[2]);
      float d2 = vdot(p, p);
      if (d2 < best) { best = d2; l[0] = tl[0]; l[1] = tl[1]; l[2] = tl[2]; l[3] = 0.0f;
        used = (tu & 1) | (tu & 2) | (tu & 4); }
    }
    if (o1) {
      closest_tri(a, c, d, tl, &tu);
      v3 p = vmadd(vmadd(vscale(a, tl[0]), c, tl[1]), d, tl[2]);
      float d2 = vdot(p, p);
      if (d2 < best) { best = d2; l[0] = tl[0]; l[1] = 0.0f; l[2] = tl[1]; l[3] = tl[2];
        used = (tu & 1) | ((tu & 2) << 1) | ((tu & 4) << 1); }
    }
    if (o2) {
      closest_tri(a, d, b, tl, &tu);
      v3 p = vmadd(vmadd(vscale(a, tl[0]), d, tl[1]), b, tl[2]);
      float d2 = vdot(p, p);
      if (d2 < best) { best = d2; l[0] = tl[0]; l[1] = tl[2]; l[2] = 0.0f; l[3] = tl[1];
        used = (tu & 1) | ((tu & 2) << 2) | ((tu & 4) >> 1); }
    }
    if (o3) {
      closest_tri(b, d, c, tl, &tu);
      v3 p = vmadd(vmadd(vscale(b, tl[0]), d, tl[1]), c, tl[2]);
      float d2 = vdot(p, p);
      if (d2 < best) { best = d2; l[0] = 0.0f; l[1] = tl[0]; l[2] = tl[2]; l[3] = tl[1];
        used = ((tu & 1) << 1) | ((tu & 2) << 2) | (tu & 4); }
    }
  }
  /* compact */
  int m = 0;
  v3 v = V(0.0f, 0.0f, 0.0f);
  for (int i = 0; i < s->n; ++i) {
    if (used & (1 << i)) {
      s->w[m] = s->w[i]; s->p[m] = s->p[i]; s->q[m] = s->q[i];
      s->ia[m] = s->ia[i]; s->ib[m] = s->ib[i];
      lam[m] = l[i];
      v = vmadd(v, s->w[m], lam[m]);
      ++m;
    }
  }
  s->n = m;
  *vout = v;
  return 1;
}

static inline int support_max(const v3* P, int n, v3 d) {
  int best = 0;
  float bd = vdot(P[0], d);
  for (int k = 1; k < n; ++k) {
    float t = vdot(P[k], d);
    if (t > bd) { bd = t; best = k; }
  }
  return best;
}

/* GJK distance between two world-space vertex clouds (convex hulls).
 * returns 0: farther than maxdist (no contact); 1: pa/pb/n/dist valid; 2: hulls overlap */
static void simplex_store(const simplex_t* s, int* sn, int* sia, int* sib) {
  *sn = s->n;
  for (int k = 0; k < 4; ++k) { sia[k] = k < s->n ? s->ia[k] : 0; sib[k] = k < s->n ? s->ib[k] : 0; }
}

/* GJK with a cached simplex: the vertex pairs that spanned the closest feature last time are re-evaluated at
 * the current poses and used as the starting simplex (a resting contact then converges in one iteration);
 * if that simplex is degenerate or encloses the origin the search restarts from the cached axis. */
static int gjk_distance(const v3* VA, int na, const v3* VB, int nb, v3* axis, int* sn, int* sia, int* sib,
                        float maxdist, v3* pa, v3* pb, v3* nrm, float* dist) {
  simplex_t s; s.n = 0;
  float lam[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  v3 v = *axis;
  float sqd = 1e30f;
  int warm = 0;
  if (*sn > 0) {
    s.n = *sn;
    for (int k = 0; k < s.n; ++k) {
      s.ia[k] = sia[k]; s.ib[k] = sib[k];
      s.p[k] = VA[sia[k]]; s.q[k] = VB[sib[k]];
      s.w[k] = vsub(s.p[k], s.q[k]);
    }
    v3 nv;
    int rc = simplex_closest(&s, lam, &nv);
    float nsq = vdot(nv, nv);
    if (rc == 1 && !(nsq < 1e-10f)) { v = nv; sqd = nsq; warm = 1; }
    else { s.n = 0; lam[0] = 0.0f; lam[1] = 0.0f; lam[2] = 0.0f; lam[3] = 0.0f; }
  }
  if (!(vdot(v, v) > 1e-20f)) v = V(0.0f, 0.0f, 1.0f);
  for (int it = 0; it < GJK_MAXIT; ++it) {
    int ia = support_max(VA, na, vneg(v));
    int ib = support_max(VB, nb, v);
    v3 w = vsub(VA[ia], VB[ib]);
    float delta = vdot(v, w);
    if (it > 0 || warm) {
      if (delta > 0.0f && delta * delta > sqd * (maxdist * maxdist)) { simplex_store(&s, sn, sia, sib); return 0; }
      int dup = 0;
      for (int k = 0; k < s.n; ++k) dup |= (s.ia[k] == ia && s.ib[k] == ib);
      if (dup) break;
      if (sqd - delta <= sqd * 1e-6f) break;
    }
    simplex_t bak = s;
    float blam[4] = {lam[0], lam[1], lam[2], lam[3]};
    s.w[s.n] = w; s.p[s.n] = VA[ia]; s.q[s.n] = VB[ib]; s.ia[s.n] = ia; s.ib[s.n] = ib; s.n++;
    v3 nv;
    int rc = simplex_closest(&s, lam, &nv);
    if (rc == 2) { *sn = 0; return 2; }
    if (rc == 0) {
      if (it == 0 && !warm) { *sn = 0; return 0; }
      s = bak; lam[0] = blam[0]; lam[1] = blam[1]; lam[2] = blam[2]; lam[3] = blam[3];
      break;
    }
    float nsq = vdot(nv, nv);
    if (nsq < 1e-10f) { *sn = 0; return 2; }
    int stall = (it > 0 || warm) && (sqd - nsq <= 1.1920929e-7f * sqd);
    v = nv; sqd = nsq;
    if (stall) break;
  }
  simplex_store(&s, sn, sia, sib);
  v3 A = V(0.0f, 0.0f, 0.0f), B = V(0.0f, 0.0f, 0.0f);
  for (int k = 0; k < s.n; ++k) {
    A = vmadd(A, s.p[k], lam[k]);
    B = vmadd(B, s.q[k], lam[k]);
  }
  float d = sqrtf(sqd);
  if (d > maxdist) { *axis = v; return 0; }
  *pa = A; *pb = B; *dist = d;
  *nrm = vscale(v, 1.0f / d);
  *axis = v;
  return 1;
}

/* Face-normal SAT for overlapping hulls: least-penetration face axis of either body. */
static void sat_faces(const mesh_t* MA, const v3* VA, const mesh_t* MB, const v3* VB, v3* pa, v3* pb,
                      v3* nrm, float* dist) {
  float best = -1e30f; int btype = 0, bvert = 0; v3 bn = V(0.0f, 0.0f, 1.0f);
  for (int pass = 0; pass < 2; ++pass) {
    const mesh_t* MF = pass ? MB : MA; const v3* VF = pass ? VB : VA;
    const v3* VO = pass ? VA : VB; int no = pass ? MA->nv : MB->nv;
    for (int t = 0; t < MF->nt; ++t) {
      v3 a = VF[MF->tri[t][0]], b = VF[MF->tri[t][1]], c = VF[MF->tri[t][2]];
      v3 n = vcross(vsub(b, a), vsub(c, a));
      float l2 = vdot(n, n);
      if (l2 < 1e-20f) continue;
      n = vscale(n, 1.0f / sqrtf(l2));
      float smin = 1e30f; int kmin = 0;
      for (int k = 0; k < no; ++k) {
        float sd = vdot(n, vsub(VO[k], a));
        if (sd < smin) { smin = sd; kmin = k; }
      }
      if (smin > best) { best = smin; btype = pass; bvert = kmin; bn = n; }
    }
  }
  if (btype == 0) { /* face of A, deepest vertex of B; B->A normal is -face normal */
    *nrm = vneg(bn); *pb = VB[bvert]; *pa = vmadd(VB[bvert], bn, -best);
  } else {          /* face of B, deepest vertex of A */
    *nrm = bn; *pa = VA[bvert]; *pb = vmadd(VA[bvert], bn, -best);
  }
  *dist = best;
}

/* ---- persistent manifold (Bullet btPersistentManifold restated) */
static void manifold_refresh(manifold_t* m, v3 xa, const m3* Ra, v3 xb, const m3* Rb, float thr) {
  for (int i = m->np - 1; i >= 0; --i) {
    mpoint_t* p = &m->p[i];
    v3 wa = mmul_add(Ra, p->la, xa);
    v3 wb = mmul_add(Rb, p->lb, xb);
    float d = vdot(vsub(wa, wb), p->n);
    int drop = d > thr;
    if (!drop) {
      v3 proj = vmadd(wa, p->n, -d);
      v3 t = vsub(wb, proj);
      drop = vdot(t, t) > thr * thr;
    }
    if (drop) { m->p[i] = m->p[m->np - 1]; m->np--; }
    else p->dist = d;
  }
}

static int manifold_sort_replace(const manifold_t* m, const mpoint_t* np_) {
  int deep = -1; float maxpen = np_->dist;
  for (int i = 0; i < 4; ++i) if (m->p[i].dist < maxpen) { deep = i; maxpen = m->p[i].dist; }
  float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (deep != 0) { v3 a = vsub(np_->la, m->p[1].la), b = vsub(m->p[3].la, m->p[2].la); v3 c = vcross(a, b); res[0] = vdot(c, c); }
  if (deep != 1) { v3 a = vsub(np_->la, m->p[0].la), b = vsub(m->p[3].la, m->p[2].la); v3 c = vcross(a, b); res[1] = vdot(c, c); }
  if (deep != 2) { v3 a = vsub(np_->la, m->p[0].la), b = vsub(m->p[3].la, m->p[1].la); v3 c = vcross(a, b); res[2] = vdot(c, c); }
  if (deep != 3) { v3 a = vsub(np_->la, m->p[0].la), b = vsub(m->p[2].la, m->p[1].la); v3 c = vcross(a, b); res[3] = vdot(c, c); }
  int best = 0;
  for (int i = 1; i < 4; ++i) if (res[i] > res[best]) best = i;
  return best;
}

static void manifold_add(manifold_t* m, const mpoint_t* np_, float thr) {
  float shortest = thr * thr; int near_i = -1;
  for (int i = 0; i < m->np; ++i) {
    v3 d = vsub(m->p[i].la, np_->la);
    float d2 = vdot(d, d);
    if (d2 < shortest) { shortest = d2; near_i = i; }
  }
  if (near_i >= 0) { /* replace geometry, keep the cached impulses */
    mpoint_t* p = &m->p[near_i];
    p->la = np_->la; p->lb = np_->lb; p->n = np_->n; p->dist = np_->dist;
  } else if (m->np < 4) {
    m->p[m->np++] = *np_;
  } else {
    m->p[manifold_sort_replace(m, np_)] = *np_;
  }
}

/* ---- one sub-step (pb.stepSimulation restated; see DESIGN.md "settle solver") */
static inline int pair_id(int i, int j) { return j * (j - 1) / 2 + i; } /* i < j */

static void derive_bodies(const struct srlo_env* e, env_t* s) {
  for (int b = 0; b < s->nb; ++b) {
    const mesh_t* M = &e->mesh[s->mesh[b]];
    s->R[b] = quat_to_mat(s->q[b]);
    s->Iw[b] = inv_inertia_world(&s->R[b], M->inv_inertia);
    v3 lo = V(1e30f, 1e30f, 1e30f), hi = V(-1e30f, -1e30f, -1e30f);
    for (int k = 0; k < M->nv; ++k) {
      v3 a = mmul_add(&s->R[b], M->v[k], s->x[b]);
      s->wv[b][k] = a;
      lo = V(fminf(lo.x, a.x), fminf(lo.y, a.y), fminf(lo.z, a.z));
      hi = V(fmaxf(hi.x, a.x), fmaxf(hi.y, a.y), fmaxf(hi.z, a.z));
    }
    float ex = e->c.collision_margin + 0.01f * M->radius;
    s->amin[b] = V(lo.x - ex, lo.y - ex, lo.z - ex);
    s->amax[b] = V(hi.x + ex, hi.y + ex, hi.z + ex);
  }
}

/* ground manifold: the deepest vertices within the breaking threshold, at most 4 — the size of Bullet's persistent
 * manifold (MANIFOLD_CACHE_SIZE); round 1 kept 8, which doubled the ground phase of every solver sweep for the same
 * sub-step and sweep counts.  (Choosing the four by extent — deepest, farthest from it, farthest to either side of that
 * line — was tried: it can leave out a vertex that carries the rock, and such rocks never come to rest.) */
#define GMAXP 4
static void ground_manifold(const struct srlo_env* e, env_t* s, int b) {
  const mesh_t* M = &e->mesh[s->mesh[b]];
  float m = e->c.collision_margin, thr = 0.02f * M->radius;
  gmanifold_t old = s->gm[b];
  gmanifold_t* g = &s->gm[b];
  const v3* W = s->wv[b];
  /* selection: repeatedly take the deepest not-yet-taken vertex with dist < thr (lowest index on ties) */
  int ns = 0;
  float last_d = -1e30f; int last_k = -1;
  while (ns < GMAXP) {
    int best = -1; float bd = thr;
    for (int k = 0; k < M->nv; ++k) {
      float d = W[k].z - m;
      /* strictly after (last_d, last_k) in (dist, index) order */
      if (!(d > last_d || (d == last_d && k > last_k))) continue;
      if (d < bd) { bd = d; best = k; }
    }
    if (best < 0) break;
    g->vid[ns] = best; g->dist[ns] = bd;
    g->in[ns] = 0.0f; g->it1[ns] = 0.0f; g->it2[ns] = 0.0f;
    for (int j = 0; j < old.np; ++j)
      if (old.vid[j] == best) { g->in[ns] = old.in[j]; g->it1[ns] = old.it1[j]; g->it2[ns] = old.it2[j]; }
    last_d = bd; last_k = best;
    ++ns;
  }
  g->np = ns;
}

static void update_slots(const struct srlo_env* e, env_t* s) {
  (void)e;
  /* broadphase over all pairs in ascending pair id; slot bookkeeping */
  int changed = 0;
  for (int j = 1; j < s->nb; ++j)
    for (int i = 0; i < j; ++i) {
      int pid = pair_id(i, j);
      int ov = s->amin[i].x <= s->amax[j].x && s->amin[j].x <= s->amax[i].x &&
               s->amin[i].y <= s->amax[j].y && s->amin[j].y <= s->amax[i].y &&
               s->amin[i].z <= s->amax[j].z && s->amin[j].z <= s->amax[i].z;
      int sl = s->slot_of_pair[pid];
      if (!ov && sl >= 0) { s->slot_of_pair[pid] = -1; s->pair_of_slot[sl] = -1; changed = 1; }
    }
  for (int j = 1; j < s->nb; ++j)
    for (int i = 0; i < j; ++i) {
      int pid = pair_id(i, j);
      int ov = s->amin[i].x <= s->amax[j].x && s->amin[j].x <= s->amax[i].x &&
               s->amin[i].y <= s->amax[j].y && s->amin[j].y <= s->amax[i].y &&
               s->amin[i].z <= s->amax[j].z && s->amin[j].z <= s->amax[i].z;
      if (ov && s->slot_of_pair[pid] < 0) {
        int sl = -1;
        const int cap = e->c.episode_length <= 16 ? 64 : 128;   /* manifold slots per env, as the kernels lay them out */
        for (int k = 0; k < MAXSLOT && k < cap; ++k) if (s->pair_of_slot[k] < 0) { sl = k; break; }
        if (sl < 0) { s->status |= SRL_ST_PAIR_OVERFLOW; continue; }
        s->slot_of_pair[pid] = (int16_t)sl; s->pair_of_slot[sl] = (int16_t)pid;
        s->slot_a[sl] = (uint8_t)i; s->slot_b[sl] = (uint8_t)j;
        s->man[sl].np = 0;
        s->man[sl].sn = 0;
        s->man[sl].axis = vsub(s->x[i], s->x[j]);
        changed = 1;
      }
    }
  if (changed || s->ncolour < 0) {
    /* greedy colouring in slot order: slots of one colour share no body */
    uint64_t used[MAXB];
    for (int b = 0; b < MAXB; ++b) used[b] = 0;
    int nc = 0;
    for (int sl = 0; sl < MAXSLOT; ++sl) {
      if (s->pair_of_slot[sl] < 0) continue;
      uint64_t u = used[s->slot_a[sl]] | used[s->slot_b[sl]];
      int c = 0;
      while (u & ((uint64_t)1 << c)) ++c;
      s->colour[sl] = (uint8_t)c;
      used[s->slot_a[sl]] |= (uint64_t)1 << c;
      used[s->slot_b[sl]] |= (uint64_t)1 << c;
      if (c + 1 > nc) nc = c + 1;
    }
    s->ncolour = nc;
  }
}

static void narrowphase_slot(const struct srlo_env* e, env_t* s, int sl) {
  int a = s->slot_a[sl], b = s->slot_b[sl];
  const mesh_t* MA = &e->mesh[s->mesh[a]];
  const mesh_t* MB = &e->mesh[s->mesh[b]];
  manifold_t* m = &s->man[sl];
  float mg = e->c.collision_margin;
  float thr = 0.02f * fminf(MA->radius, MB->radius);
  manifold_refresh(m, s->x[a], &s->R[a], s->x[b], &s->R[b], thr);
  v3 pa, pb, n; float d;
  int rc = gjk_distance(s->wv[a], MA->nv, s->wv[b], MB->nv, &m->axis, &m->sn, m->sia, m->sib, (mg + mg) + thr, &pa, &pb, &n, &d);
  if (rc == 2) { sat_faces(MA, s->wv[a], MB, s->wv[b], &pa, &pb, &n, &d); rc = 1; }
  if (rc == 1) {
    float dist = d - (mg + mg);
    if (dist < thr) {
      mpoint_t np_;
      v3 sa = vmadd(pa, n, -mg);
      v3 sb = vmadd(pb, n, mg);
      np_.la = mtmul(&s->R[a], vsub(sa, s->x[a]));
      np_.lb = mtmul(&s->R[b], vsub(sb, s->x[b]));
      np_.n = n; np_.dist = dist; np_.in = 0.0f; np_.it1 = 0.0f; np_.it2 = 0.0f;
      manifold_add(m, &np_, thr);
    }
  }
}

/* one solver row: direction d at arms ra (body A) / rb (body B) */
typedef struct { v3 va, wa, vb, wb; } vel4;

/* The clamp of an accumulated impulse to [lo, hi] is the median of the three values, as the GPU instruction the kernel
 * uses defines it for non-NaN inputs (V_MED3_F32: if MAX3 == S0 then MAX(S1, S2), else if MAX3 == S1 then MAX(S0, S2),
 * else MAX(S0, S1), with MAX ordering -0 below +0).  For lo <= hi this is the usual clamp; the restatement pins which
 * zero comes back when the value equals a bound of the other sign. */
static inline float fmax_sz(float a, float b) { if (a > b) return a; if (b > a) return b; return signbit(a) ? b : a; }
static inline float med3f(float a, float b, float c) {
  float m = fmax_sz(fmax_sz(a, b), c);
  if (m == a) return fmax_sz(b, c);
  if (m == b) return fmax_sz(a, c);
  return fmax_sz(a, b);
}

static inline float row_solve(v3 d, v3 ra, v3 rb, float ima, const m3* Ia, float imb, const m3* Ib,
                              vel4* u, float target, float* acc, float lo, float hi, int has_b) {
  v3 ca = vcross(ra, d);
  v3 aa = mmul(Ia, ca);
  v3 cb = V(0.0f, 0.0f, 0.0f), ab = V(0.0f, 0.0f, 0.0f);
  float k = ima + vdot(vcross(aa, ra), d);
  float vrel = vdot(d, u->va) + vdot(ca, u->wa);
  if (has_b) {
    cb = vcross(rb, d);
    ab = mmul(Ib, cb);
    k = k + (imb + vdot(vcross(ab, rb), d));
    vrel = vrel - (vdot(d, u->vb) + vdot(cb, u->wb));
  }
  float rk = 1.0f / k;            /* reciprocal effective mass, as the kernel precomputes it per row */
  /* (target - vrel) rk as one fused step, and the linear impulses as (d m^-1) dl: each form keeps the solver's
   * dependency chain one operation shorter than (target - vrel) * rk and d (m^-1 dl); the kernel evaluates the same */
  float dl = fmaf(-vrel, rk, target * rk);
  float na = med3f(*acc + dl, lo, hi);   /* clamp to [lo, hi] */
  dl = na - *acc;
  *acc = na;
  u->va = vmadd(u->va, vscale(d, ima), dl);
  u->wa = vmadd(u->wa, aa, dl);
  if (has_b) {
    u->vb = vmadd(u->vb, vscale(d, -imb), dl);
    u->wb = vmadd(u->wb, ab, -dl);
  }
  /* the row's residual as Bullet's resolveSingleConstraintRow* returns it: the applied (clamped) delta impulse times
   * the effective-mass denominator (deltaImpulse / m_jacDiagABInv), i.e. the velocity change it stands for */
  return fabsf(dl * k);
}

static inline void row_apply(v3 d, v3 ra, v3 rb, float ima, const m3* Ia, float imb, const m3* Ib,
                             vel4* u, float imp, int has_b) {
  v3 aa = mmul(Ia, vcross(ra, d));
  u->va = vmadd(u->va, vscale(d, ima), imp);
  u->wa = vmadd(u->wa, aa, imp);
  if (has_b) {
    v3 ab = mmul(Ib, vcross(rb, d));
    u->vb = vmadd(u->vb, vscale(d, -imb), imp);
    u->wb = vmadd(u->wb, ab, -imp);
  }
}

/* Ground rows.  The ground's contact normal is +z and its tangents (plane_space of +z) are -y and +x, so a ground row's
 * direction is sgn * e_axis: its relative velocity needs one component of the linear velocity and its impulse changes
 * that component only — stated as such (not as products with the zeros of d) so that the kernel and this file evaluate
 * the same, shorter expressions: vrel = ca . w + sgn v[axis] as one fused chain, v[axis] += (sgn m^-1) dl.  The row
 * constants (ca = ra x d, aa = I ca, k) are those of the general row. */
#define ROW_AXIS_FUNCS(NAME, COMP)                                                                                    \
static inline float row_solve_##NAME(float sgn, v3 d, v3 ra, float ima, const m3* Ia, vel4* u, float target,         \
                                     float* acc, float lo, float hi) {                                               \
  v3 ca = vcross(ra, d);                                                                                             \
  v3 aa = mmul(Ia, ca);                                                                                              \
  float k = ima + vdot(vcross(aa, ra), d);                                                                           \
  float vrel = fmaf(ca.x, u->wa.x, fmaf(ca.y, u->wa.y, fmaf(ca.z, u->wa.z, sgn * u->va.COMP)));                       \
  float rk = 1.0f / k;                                                                                               \
  float dl = fmaf(-vrel, rk, target * rk);                                                                           \
  float na = med3f(*acc + dl, lo, hi);                                                                               \
  dl = na - *acc;                                                                                                    \
  *acc = na;                                                                                                         \
  u->va.COMP = fmaf(sgn * ima, dl, u->va.COMP);                                                                      \
  u->wa = vmadd(u->wa, aa, dl);                                                                                      \
  return fabsf(dl * k);                                                                                              \
}                                                                                                                    \
static inline void row_apply_##NAME(float sgn, v3 d, v3 ra, float ima, const m3* Ia, vel4* u, float imp) {           \
  v3 aa = mmul(Ia, vcross(ra, d));                                                                                   \
  u->va.COMP = fmaf(sgn * ima, imp, u->va.COMP);                                                                     \
  u->wa = vmadd(u->wa, aa, imp);                                                                                     \
}
ROW_AXIS_FUNCS(x, x)
ROW_AXIS_FUNCS(y, y)
ROW_AXIS_FUNCS(z, z)

static inline float contact_target(const struct srlo_env* e, float dist) {
  float inv_dt = 1.0f / e->c.sim_time_step;
  /* Bullet setupContactConstraint restated: penetration = distance + m_linearSlop; separated points may close the gap
   * in one step, penetrating points are pushed out with erp */
  float pen = dist + e->c.linear_slop;
  return pen > 0.0f ? -(pen * inv_dt) : -((pen * e->c.erp) * inv_dt);
}

static float solve_ground(const struct srlo_env* e, env_t* s, int b, int warm) {
  gmanifold_t* g = &s->gm[b];
  float res = 0.0f;
  if (g->np == 0) return res;
  const mesh_t* M = &e->mesh[s->mesh[b]];
  float mu = e->c.friction_rock * e->c.friction_ground;
  vel4 u; u.va = s->v[b]; u.wa = s->w[b]; u.vb = V(0, 0, 0); u.wb = V(0, 0, 0);
  v3 n = V(0.0f, 0.0f, 1.0f), t1, t2;
  plane_space(n, &t1, &t2);
  for (int i = 0; i < g->np; ++i) {
    v3 pw = s->wv[b][g->vid[i]];
    v3 ra = vsub(V(pw.x, pw.y, pw.z - e->c.collision_margin), s->x[b]);
    if (warm) {
      g->in[i] = g->in[i] * e->c.warmstart; g->it1[i] = g->it1[i] * e->c.warmstart; g->it2[i] = g->it2[i] * e->c.warmstart;
      row_apply_z(1.0f, n, ra, M->inv_mass, &s->Iw[b], &u, g->in[i]);
      row_apply_y(-1.0f, t1, ra, M->inv_mass, &s->Iw[b], &u, g->it1[i]);
      row_apply_x(1.0f, t2, ra, M->inv_mass, &s->Iw[b], &u, g->it2[i]);
    } else {
      res = fmaxf(res, row_solve_z(1.0f, n, ra, M->inv_mass, &s->Iw[b], &u, contact_target(e, g->dist[i]), &g->in[i], 0.0f, 1e30f));
      float lim = mu * g->in[i];
      res = fmaxf(res, row_solve_y(-1.0f, t1, ra, M->inv_mass, &s->Iw[b], &u, 0.0f, &g->it1[i], -lim, lim));
      res = fmaxf(res, row_solve_x(1.0f, t2, ra, M->inv_mass, &s->Iw[b], &u, 0.0f, &g->it2[i], -lim, lim));
    }
  }
  s->v[b] = u.va; s->w[b] = u.wa;
  return res;
}

static float solve_slot(const struct srlo_env* e, env_t* s, int sl, int warm) {
  manifold_t* m = &s->man[sl];
  float res = 0.0f;
  if (m->np == 0) return res;
  int a = s->slot_a[sl], b = s->slot_b[sl];
  const mesh_t* MA = &e->mesh[s->mesh[a]];
  const mesh_t* MB = &e->mesh[s->mesh[b]];
  float mu = e->c.friction_rock * e->c.friction_rock;
  vel4 u; u.va = s->v[a]; u.wa = s->w[a]; u.vb = s->v[b]; u.wb = s->w[b];
  for (int i = 0; i < m->np; ++i) {
    mpoint_t* p = &m->p[i];
    v3 ra = mmul(&s->R[a], p->la);
    v3 rb = mmul(&s->R[b], p->lb);
    v3 t1, t2;
    plane_space(p->n, &t1, &t2);
    if (warm) {
      p->in = p->in * e->c.warmstart; p->it1 = p->it1 * e->c.warmstart; p->it2 = p->it2 * e->c.warmstart;
      row_apply(p->n, ra, rb, MA->inv_mass, &s->Iw[a], MB->inv_mass, &s->Iw[b], &u, p->in, 1);
      row_apply(t1, ra, rb, MA->inv_mass, &s->Iw[a], MB->inv_mass, &s->Iw[b], &u, p->it1, 1);
      row_apply(t2, ra, rb, MA->inv_mass, &s->Iw[a], MB->inv_mass, &s->Iw[b], &u, p->it2, 1);
    } else {
      res = fmaxf(res, row_solve(p->n, ra, rb, MA->inv_mass, &s->Iw[a], MB->inv_mass, &s->Iw[b], &u, contact_target(e, p->dist), &p->in, 0.0f, 1e30f, 1));
      float lim = mu * p->in;
      res = fmaxf(res, row_solve(t1, ra, rb, MA->inv_mass, &s->Iw[a], MB->inv_mass, &s->Iw[b], &u, 0.0f, &p->it1, -lim, lim, 1));
      res = fmaxf(res, row_solve(t2, ra, rb, MA->inv_mass, &s->Iw[a], MB->inv_mass, &s->Iw[b], &u, 0.0f, &p->it2, -lim, lim, 1));
    }
  }
  s->v[a] = u.va; s->w[a] = u.wa; s->v[b] = u.vb; s->w[b] = u.wb;
  return res;
}

/* one Gauss-Seidel sweep over every contact row of the env; returns the largest row residual of the sweep (the
 * maximum is order-independent, so the kernel's parallel sweep reports the same bits) */
static float solver_sweep(const struct srlo_env* e, env_t* s, int warm) {
  float res = 0.0f;
  for (int b = 0; b < s->nb; ++b) res = fmaxf(res, solve_ground(e, s, b, warm));
  for (int c = 0; c < s->ncolour; ++c)
    for (int sl = 0; sl < MAXSLOT; ++sl)
      if (s->pair_of_slot[sl] >= 0 && s->colour[sl] == c) res = fmaxf(res, solve_slot(e, s, sl, warm));
  return res;
}

static void substep(const struct srlo_env* e, env_t* s) {
  float dt = e->c.sim_time_step;
  /* damping then gravity (btRigidBody::applyDamping, then the external-force impulse) */
  for (int b = 0; b < s->nb; ++b) {
    s->v[b] = vscale(s->v[b], e->lin_damp);
    s->w[b] = vscale(s->w[b], e->ang_damp);
    s->v[b].z = s->v[b].z - e->c.gravity * dt;
  }
  derive_bodies(e, s);
  for (int b = 0; b < s->nb; ++b) ground_manifold(e, s, b);
  update_slots(e, s);
  for (int sl = 0; sl < MAXSLOT; ++sl)
    if (s->pair_of_slot[sl] >= 0) narrowphase_slot(e, s, sl);
  solver_sweep(e, s, 1);
  /* btSequentialImpulseConstraintSolver::solveGroupCacheFriendlyIterations restated: at most numIterations sweeps,
   * ended early once the largest squared row residual of a sweep is <= m_leastSquaresResidualThreshold */
  for (int it = 0; it < e->c.solver_iterations; ++it) {
    float res = solver_sweep(e, s, 0);
    s->sweeps++;
    if (res * res <= e->c.residual_threshold) break;
  }
  /* integrate */
  for (int b = 0; b < s->nb; ++b) {
    s->x[b] = vmadd(s->x[b], s->v[b], dt);
    q4 q = s->q[b]; v3 w = s->w[b];
    float hx = 0.5f * dt;
    q4 dq;
    dq.x = hx * ((w.x * q.w + w.y * q.z) - w.z * q.y);
    dq.y = hx * ((w.y * q.w + w.z * q.x) - w.x * q.z);
    dq.z = hx * ((w.z * q.w + w.x * q.y) - w.y * q.x);
    dq.w = hx * (-((w.x * q.x + w.y * q.y) + w.z * q.z));
    q.x += dq.x; q.y += dq.y; q.z += dq.z; q.w += dq.w;
    float inv = 1.0f / sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
    q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv;
    s->q[b] = q;
  }
}

/* ------------------------------------------------------------------ Simulator.step (simulator.py:190-258)
 * The loop is written over the WORLD the reference drives — one entry per pybullet call site of simulator.py — so that the
 * same code serves the oracle's env (its own rigid-body step behind `step`) and a scripted world
 * (srlo_sim_step_scripted, below): tests/test_simulator_golden.py compares the sequence of calls, the counters and the
 * exits of this function with those of the reference's own simulator.py run over a recording pybullet placeholder
 * (tests/golden/make_simulator_golden.py). */
typedef struct sim_world {
  void* ctx;
  int (*has_new)(void* ctx);                 /* `if self._new:` (simulator.py:312) */
  void (*place)(void* ctx);                  /* resetBasePositionAndOrientation(new, ...); objects.append (simulator.py:313-318) */
  void (*step)(void* ctx);                   /* stepSimulation (simulator.py:219, :240, :320) */
  void (*zero_newest)(void* ctx);            /* resetBaseVelocity(objects[-1], 0, 0) (simulator.py:214-218) */
  int (*newest_contacts)(void* ctx);         /* len(getContactPoints(objects[-1])) (simulator.py:340) */
  int (*n_objects)(void* ctx);               /* len(objects) */
  float (*linear_speed)(void* ctx, int b);   /* norm(getBaseVelocity(objects[b])[0]) (simulator.py:332-333) */
  void (*store_place_pose)(void* ctx);       /* place_poses.append(getBasePositionAndOrientation(objects[-1])) (simulator.py:227) */
} sim_world;

/* simulator.py:322-335 (velocity criterion, newest body first; angular velocity is ignored) */
static int sim_stop(const sim_world* w, float thr) {
  for (int b = w->n_objects(w->ctx) - 1; b >= 0; --b)
    if (w->linear_speed(w->ctx, b) > thr) return 0;
  return 1;
}
/* simulator.py:337-341 */
static int sim_drop(const sim_world* w, float thr) {
  return w->newest_contacts(w->ctx) >= 3 || sim_stop(w, thr);
}

/* returns 1 where the reference raises RuntimeError (simulator.py:221-224, :242-245); substeps = `Simulator.n_steps` */
static int sim_step_world(const sim_world* w, int smooth_placing, float thr, int max_substeps, int32_t* substeps) {
  int counter, diverged = 0;
  if (w->has_new(w->ctx)) {          /* _place, simulator.py:310-320 */
    w->place(w->ctx);
    w->step(w->ctx);
  }
  counter = 1;
  if (smooth_placing) {
    while (!sim_drop(w, thr)) {
      w->zero_newest(w->ctx);
      w->step(w->ctx);
      counter++;
      if (counter > max_substeps) { diverged = 1; break; }
    }
  }
  if (diverged) { substeps[0] = counter; substeps[1] = 0; return 1; }
  w->store_place_pose(w->ctx);
  substeps[0] = counter;
  while (!sim_stop(w, thr)) {
    w->step(w->ctx);
    counter++;
    if (counter > max_substeps) { diverged = 1; break; }
  }
  substeps[1] = counter - substeps[0];
  return diverged;
}

/* the oracle's env as that world */
typedef struct { const struct srlo_env* e; env_t* s; v3 pos; int oi; } env_world;
static int ew_has_new(void* c) { return ((env_world*)c)->s->pending >= 0; }
static void ew_place(void* c) {
  env_world* w = (env_world*)c;
  const struct srlo_env* e = w->e; env_t* s = w->s;
  int b = s->nb;
  const mesh_t* M = &e->mesh[s->pending];
  s->mesh[b] = s->pending;
  /* resetBasePositionAndOrientation moves the inertial (COM) frame; loadURDF placed the link frame */
  if (e->n_orient == 1) {
    s->x[b] = e->c.place_at_com ? w->pos : vadd(w->pos, M->com);
    s->q[b].x = 0.0f; s->q[b].y = 0.0f; s->q[b].z = 0.0f; s->q[b].w = 1.0f;
  } else {   /* the chosen orientation (observer.py:416-417 -> simulator.py:313) */
    m3 Ro = quat_to_mat(e->orient_q[w->oi]);
    s->x[b] = e->c.place_at_com ? w->pos : vadd(w->pos, mmul(&Ro, M->com));
    s->q[b] = e->orient_q[w->oi];
  }
  s->v[b] = V(0, 0, 0); s->w[b] = V(0, 0, 0);
  s->gm[b].np = 0;
  s->nb = b + 1;
  s->pending = -1;
}
static void ew_step(void* c) { env_world* w = (env_world*)c; substep(w->e, w->s); }
static void ew_zero_newest(void* c) { env_t* s = ((env_world*)c)->s; s->v[s->nb - 1] = V(0, 0, 0); s->w[s->nb - 1] = V(0, 0, 0); }
static int ew_newest_contacts(void* c) {
  const env_t* s = ((env_world*)c)->s;
  int b = s->nb - 1;
  int n = s->gm[b].np;
  for (int sl = 0; sl < MAXSLOT; ++sl)
    if (s->pair_of_slot[sl] >= 0 && (s->slot_a[sl] == b || s->slot_b[sl] == b)) n += s->man[sl].np;
  return n;
}
static int ew_n_objects(void* c) { return ((env_world*)c)->s->nb; }
static float ew_linear_speed(void* c, int b) { const env_t* s = ((env_world*)c)->s; return sqrtf(vdot(s->v[b], s->v[b])); }
static void ew_store_place_pose(void* c) {
  env_t* s = ((env_world*)c)->s;
  s->place_x[s->nb - 1] = s->x[s->nb - 1];
  s->place_q[s->nb - 1] = s->q[s->nb - 1];
}

static void sim_step(const struct srlo_env* e, env_t* s, v3 pos, int oi) {
  env_world ctx = {e, s, pos, oi};
  const sim_world w = {&ctx, ew_has_new, ew_place, ew_step, ew_zero_newest, ew_newest_contacts, ew_n_objects, ew_linear_speed,
                       ew_store_place_pose};
  s->sweeps = 0;
  /* the reference raises RuntimeError at the cap (simulator.py:221-224, :242-245); here the flag stays set until the next
   * reset so the caller can see which env it was */
  if (sim_step_world(&w, e->c.smooth_placing, e->c.velocity_threshold, e->max_substeps, s->substeps)) s->status |= SRL_ST_DIVERGED;
}

/* `Simulator.distances_from_place` (simulator.py:113-128) of one rock: translation |p_place - p_now| and rotation
 * 2 acos(min(w, 1)) with w the scalar part of the difference quaternion (pybullet's getDifferenceQuaternion; here the
 * quaternions' dot product, of the nearer of q and -q) */
static void distance_from_place(v3 px, q4 a, v3 x, q4 q, float* dp_out, float* do_out) {
  v3 dp = vsub(px, x);
  float dw = fabsf((a.x * q.x + a.y * q.y) + (a.z * q.z + a.w * q.w));
  *dp_out = sqrtf(vdot(dp, dp));
  *do_out = 2.0f * srlo_acosf(fminf(dw, 1.0f));
}

/* The same loop over a SCRIPTED world (the fixture entry point: tests/golden/simulator_golden.npz).  After k calls of
 * stepSimulation inside this Simulator.step, body b moves at speeds[k * nb + b] and the newest body has contacts[k]
 * contact points (k is clamped to n_rows - 1).  log receives the calls in order as codes: 1 place, 2 stepSimulation,
 * 3 resetBaseVelocity(newest, 0, 0), 4 getContactPoints(newest), 5 getBasePositionAndOrientation(newest) for the place
 * pose, 16 + b getBaseVelocity(objects[b]).  Returns the number of codes (or -1 if log_cap is too small). */
typedef struct {
  int has_new, nb, k, n_rows, nb_cols, n_log, log_cap;
  const float* speeds; const int32_t* contacts; int32_t* log;
} script_world;
static void sw_log(script_world* w, int code) { if (w->n_log < w->log_cap) w->log[w->n_log] = code; w->n_log++; }
static int sw_row(const script_world* w) { return w->k < w->n_rows ? w->k : w->n_rows - 1; }
static int sw_has_new(void* c) { return ((script_world*)c)->has_new; }
static void sw_place(void* c) { script_world* w = (script_world*)c; w->nb += 1; w->has_new = 0; sw_log(w, 1); }
static void sw_step(void* c) { script_world* w = (script_world*)c; w->k += 1; sw_log(w, 2); }
static void sw_zero(void* c) { sw_log((script_world*)c, 3); }
static int sw_contacts(void* c) { script_world* w = (script_world*)c; sw_log(w, 4); return w->contacts[sw_row(w)]; }
static int sw_n(void* c) { return ((script_world*)c)->nb; }
static float sw_speed(void* c, int b) { script_world* w = (script_world*)c; sw_log(w, 16 + b); return w->speeds[sw_row(w) * w->nb_cols + b]; }
static void sw_store(void* c) { sw_log((script_world*)c, 5); }

int srlo_sim_step_scripted(int32_t has_new, int32_t n_objects_before, int32_t smooth_placing, float velocity_threshold,
                           int32_t max_substeps, int32_t n_rows, int32_t n_cols, const float* speeds, const int32_t* contacts,
                           int32_t* substeps2, int32_t* raised, int32_t* log, int32_t log_cap) {
  script_world ctx = {has_new, n_objects_before, 0, n_rows, n_cols, 0, log_cap, speeds, contacts, log};
  const sim_world w = {&ctx, sw_has_new, sw_place, sw_step, sw_zero, sw_contacts, sw_n, sw_speed, sw_store};
  *raised = sim_step_world(&w, smooth_placing, velocity_threshold, max_substeps, substeps2);
  return ctx.n_log <= log_cap ? ctx.n_log : -1;
}

/* `Simulator.distances_from_place` of one rock from explicit poses (x, y, z, qx, qy, qz, qw) */
void srlo_distance_from_place(const float* place7, const float* now7, float* out2) {
  q4 a = {place7[3], place7[4], place7[5], place7[6]}, q = {now7[3], now7[4], now7[5], now7[6]};
  distance_from_place(V(place7[0], place7[1], place7[2]), a, V(now7[0], now7[1], now7[2]), q, &out2[0], &out2[1]);
}

/* `int(MAX_STEP_TIME / time_step)` (simulator.py:6, :46) as the library derives it from a configuration */
int32_t srlo_max_substeps(const srl_config* cfg) {
  struct srlo_env e; memset(&e, 0, sizeof e); e.c = *cfg;
  return c_max_substeps(&e.c);
}

/* ------------------------------------------------------------------ reward (rewarder.py:162-179, :261-295) */
/* `Rewarder._discount` (rewarder.py:261-269) of a translation / rotation error */
static float discount_of(const struct srlo_env* e, float perr, float oerr) {
  const srl_config* c = &e->c;
  float pmax = (float)c->object_res * e->px;     /* rewarder.py:126 */
  float omax = 3.14159265358979f;
  float disc = 1.0f;
  if (c->reward_pexp >= 0) {
    float t = perr / pmax, pw = 1.0f;
    for (int k = 0; k < c->reward_pexp; ++k) pw = pw * t;
    disc = disc * fmaxf(0.0f, 1.0f - pw);
  }
  if (c->reward_oexp >= 0) {
    float t = oerr / omax, pw = 1.0f;
    for (int k = 0; k < c->reward_oexp; ++k) pw = pw * t;
    disc = disc * fmaxf(0.0f, 1.0f - pw);
  }
  return disc;
}

/* What `Rewarder` reads of the world: the overhead map, the goal rectangle, and per rock its position and its
 * (translation, rotation) distance from the pose it was placed at (`Simulator.positions`, `.distances_from_place`). */
typedef struct {
  const float* H; const int32_t* goal; int nb;
  const float* pos;      /* [nb][3] */
  const float* dist;     /* [nb][2] = perr, oerr */
} rew_in_t;

/* the current value of one metric (rewarder.py:162-175) */
static float metric_value(const struct srlo_env* e, const rew_in_t* s, int metric) {
  const srl_config* c = &e->c;
  if (metric == SRL_METRIC_IOU || metric == SRL_METRIC_OR) {
    float inter, uni;
    srlo_iou_sums(c, s->H, s->goal, &inter, &uni);
    if (metric == SRL_METRIC_OR) return inter / ((float)(s->goal[2] * s->goal[3]) * e->goal_z);
    return inter / uni;
  }
  float r = 0.0f; int nout = 0;
  for (int b = 0; b < s->nb; ++b) {
    float fu = floorf(s->pos[3 * b] / e->px), fv = floorf(s->pos[3 * b + 1] / e->px); /* xy_to_pixel: // */
    int in = fu >= (float)s->goal[0] && fv >= (float)s->goal[1] &&
             fu < (float)(s->goal[0] + s->goal[2]) && fv < (float)(s->goal[1] + s->goal[3]);
    if (!in) { nout++; continue; }
    r = r + discount_of(e, s->dist[2 * b], s->dist[2 * b + 1]);
  }
  if (metric == SRL_METRIC_DOR) return r / (float)c->episode_length;
  return r / (float)(c->episode_length + nout);
}

/* average discount of all rocks, inside the goal or not (`_discounted(intersection=False)`, rewarder.py:149-151) */
static float average_discount(const struct srlo_env* e, const rew_in_t* s) {
  float d = 0.0f;
  for (int b = 0; b < s->nb; ++b) d = d + discount_of(e, s->dist[2 * b], s->dist[2 * b + 1]);
  return d / (float)s->nb;
}

/* Rewarder.__call__ (rewarder.py:144-160): the step's reward(s) into out[0 .. n_rewards); memory = Rewarder._memory */
static void rewarder_call(const struct srlo_env* e, const rew_in_t* s, float* memory, float* out) {
  const srl_config* c = &e->c;
  if (c->metric == SRL_METRIC_ALL) {
    for (int m = 0; m < 4; ++m) {
      float mv = metric_value(e, s, m);
      out[m] = (mv - memory[m]) * e->scale;              /* rewarder.py:176-179 */
      memory[m] = mv;
    }
  } else if (c->metric == SRL_METRIC_EVAL) {
    float mv = metric_value(e, s, SRL_METRIC_IOU);
    out[0] = (mv - memory[0]) * e->scale;
    memory[0] = mv;
    float ad = average_discount(e, s);
    out[1] = ad - memory[3];                             /* memory[-1] of a 4-entry array, not scaled */
    memory[3] = ad;
  } else {
    float mv = metric_value(e, s, c->metric);
    out[0] = (mv - memory[c->metric]) * e->scale;
    memory[c->metric] = mv;
  }
}

/* `Simulator.distances_from_place` (simulator.py:113-128) of every rock, then the rewarder on that state */
static void step_rewards(const struct srlo_env* e, env_t* s, float* out) {
  float pos[3 * MAXB], dist[2 * MAXB];
  for (int b = 0; b < s->nb; ++b) {
    pos[3 * b] = s->x[b].x; pos[3 * b + 1] = s->x[b].y; pos[3 * b + 2] = s->x[b].z;
    distance_from_place(s->place_x[b], s->place_q[b], s->x[b], s->q[b], &dist[2 * b], &dist[2 * b + 1]);
  }
  int32_t goal[4] = {s->goal[0], s->goal[1], s->goal[2], s->goal[3]};
  rew_in_t in = {s->H, goal, s->nb, pos, dist};
  rewarder_call(e, &in, s->prev_metric, out);
}

/* `Rewarder.__call__` on an explicit state (the fixture entry point: tests/golden/rewarder_golden.npz) */
int srlo_rewarder_call(const srl_config* cfg, const float* H, const int32_t* goal_rect, int32_t n_bodies,
                       const float* positions, const float* distances, float* memory, float* out) {
  struct srlo_env tmp;
  memset(&tmp, 0, sizeof tmp);
  tmp.c = *cfg;
  int rc = derive(&tmp);
  if (rc) return rc;
  rew_in_t in = {H, goal_rect, n_bodies, positions, distances};
  rewarder_call(&tmp, &in, memory, out);
  return SRL_OK;
}

/* ------------------------------------------------------------------ episode machine */
/* `StackEnv.observation` + `_return` (env.py:225-231, :171-172): stack [H, G], cast uint8(x 255 / max(max_z, omd)) in float32
 * (truncating); n_obj = number of float32 object-map pixels */
static void pack_obs_raw(const struct srlo_env* e, const float* H, const int32_t* goal, const float* O, int n_obj,
                         uint8_t* om, uint8_t* oo) {
  const srl_config* c = &e->c;
  int res = c->overhead_res;
  float den = fmaxf(c->max_z, c->object_max_dimension);            /* env.py:171-172 */
  for (int i = 0; i < res; ++i)
    for (int j = 0; j < res; ++j) {
      int in = (i >= goal[0] && i < goal[0] + goal[2] && j >= goal[1] && j < goal[1] + goal[3]);
      float g = in ? e->goal_z : 0.0f;
      om[(i * res + j) * 2 + 0] = (uint8_t)((H[i * res + j] * 255.0f) / den);
      om[(i * res + j) * 2 + 1] = (uint8_t)((g * 255.0f) / den);
    }
  for (int k = 0; k < n_obj; ++k) oo[k] = (uint8_t)((O[k] * 255.0f) / den);
}

static void pack_obs(const struct srlo_env* e, const env_t* s, uint8_t* om, uint8_t* oo) {
  const int32_t goal[4] = {s->goal[0], s->goal[1], s->goal[2], s->goal[3]};
  pack_obs_raw(e, s->H, goal, s->O, e->c.object_res * e->c.object_res * e->n_slots, om, oo);
}

/* fixture entry points (tests/golden/rewarder_golden.npz): observation packing of an explicit (H, goal, O) and the
 * action unflatten of env.py:240-241 */
int srlo_pack_observation(const srl_config* cfg, const float* H, const int32_t* goal_rect, const float* O,
                          uint8_t* obs_map, uint8_t* obs_obj) {
  struct srlo_env tmp;
  memset(&tmp, 0, sizeof tmp);
  tmp.c = *cfg;
  int rc = derive(&tmp);
  if (rc) return rc;
  pack_obs_raw(&tmp, H, goal_rect, O, cfg->object_res * cfg->object_res, obs_map, obs_obj);
  return SRL_OK;
}

/* Observer.__call__ object branch for the env's state: the pending rock (observer.py:262-277), or with ordering freedom
 * every rock still unplaced, in list order (observer.py:310-327), then empty maps up to the fixed slot count */
static void observe_objects(const struct srlo_env* e, env_t* s) {
  if (!e->c.ordering_freedom) { render_object(e, s->pending, s->O); return; }
  size_t rr = (size_t)e->c.object_res * e->c.object_res * e->n_orient;
  for (int k = 0; k < e->c.episode_length; ++k) render_object(e, k < s->list_pos ? s->ids[k] : -1, s->O + rr * k);
}

static void env_reset(struct srlo_env* e, int i) {
  env_t* s = &e->env[i];
  const srl_config* c = &e->c;
  int L = c->episode_length;
  uint32_t key = e->seed + (uint32_t)c->env_index_offset + (uint32_t)i;   /* utils.py:433 */
  s->episode += 1;
  if (s->has_script) {
    for (int k = 0; k < L; ++k) s->ids[k] = s->script_ids[k];
    for (int k = 0; k < 4; ++k) s->goal[k] = s->script_goal[k];
    s->has_script = 0;
  } else {
    /* env.py:268-272: L mesh files without replacement (with, if the pool is smaller) */
    uint32_t draw = 0;
    for (int k = 0; k < L; ++k) {
      for (;;) {
        int id = (int)rng_below(srlo_rng(key, s->episode, STREAM_MESH, draw++), (uint32_t)e->n_mesh);
        int dup = 0;
        if (e->n_mesh >= L) for (int j = 0; j < k; ++j) dup |= (s->ids[j] == id);
        if (!dup) { s->ids[k] = id; break; }
      }
    }
    srlo_goal_from_rng(c, key, s->episode, s->goal);
  }
  /* Simulator.reset: empty world + first rock pending (simulator.py:156-188) */
  s->nb = 0;
  for (int k = 0; k < NPAIR; ++k) s->slot_of_pair[k] = -1;
  for (int k = 0; k < MAXSLOT; ++k) s->pair_of_slot[k] = -1;
  s->ncolour = -1;
  s->pending = s->ids[0];
  s->list_pos = c->ordering_freedom ? L : 1;   /* with ordering freedom: the number of rocks still unplaced, ids[0 .. list_pos) */
  for (int k = 0; k < 4; ++k) s->prev_metric[k] = 0.0f;                   /* rewarder.py:191-194 */
  s->substeps[0] = 0; s->substeps[1] = 0;
  s->status = 0;
  render_heightmap(e, 0, s->mesh, s->x, s->q, s->H);
  observe_objects(e, s);
  s->done = 0;
}

int srlo_reset(srlo_env* e, uint8_t* obs_map, uint8_t* obs_obj) {
  if (!e->mesh) return fail(SRL_ENOMESH, "load meshes first");
  const srl_config* c = &e->c;
  size_t nm = (size_t)c->overhead_res * c->overhead_res * 2, no = (size_t)c->object_res * c->object_res * e->n_slots;
  for (int i = 0; i < c->n_envs; ++i) {
    env_reset(e, i);
    pack_obs(e, &e->env[i], obs_map + nm * i, obs_obj + no * i);
  }
  return SRL_OK;
}

int srlo_step(srlo_env* e, const int64_t* action, uint8_t* obs_map, uint8_t* obs_obj, float* reward,
              uint8_t* done) {
  if (!e->mesh) return fail(SRL_ENOMESH, "load meshes first");
  const srl_config* c = &e->c;
  size_t nm = (size_t)c->overhead_res * c->overhead_res * 2, no = (size_t)c->object_res * c->object_res * e->n_slots;
  int rc = SRL_OK;
  const int K = c->metric == SRL_METRIC_ALL ? 4 : c->metric == SRL_METRIC_EVAL ? 2 : 1;   /* rewards per env */
  for (int i = 0; i < c->n_envs; ++i) {
    env_t* s = &e->env[i];
    for (int k = 0; k < K; ++k) reward[(size_t)i * K + k] = 0.0f;
    if (action[i] == (int64_t)SRL_ACTION_HOLD) {         /* the env sits this call out (srl_types.h) */
      done[i] = 0;
      pack_obs(e, s, obs_map + nm * i, obs_obj + no * i);
      continue;
    }
    if (s->done) {                                       /* env.py:235-236 */
      env_reset(e, i);
      done[i] = 0;
      pack_obs(e, s, obs_map + nm * i, obs_obj + no * i);
      continue;
    }
    int64_t a = action[i];
    int oi = 0, slot = 0;                                /* TestStackEnv: (observation index, pixel), env.py:485-494 */
    if (e->n_slots > 1 && a >= 0) {
      /* the index addresses the object maps on show: n_orient of the pending rock, or n_orient of every unplaced rock */
      int nvalid = c->ordering_freedom ? s->list_pos * e->n_orient : e->n_orient;
      slot = (int)(a / (int64_t)e->A);
      a = slot < nvalid ? a % (int64_t)e->A : -1;
      oi = slot % e->n_orient;
    }
    if (a < 0 || a >= (int64_t)e->A) {                   /* env.py:238 */
      s->status |= SRL_ST_BAD_ACTION;
      done[i] = 0;
      pack_obs(e, s, obs_map + nm * i, obs_obj + no * i);
      rc = SRL_EINVAL_ACTION;
      continue;
    }
    s->status &= ~SRL_ST_BAD_ACTION;
    int u = (int)(a / e->AW), v = (int)(a % e->AW);      /* env.py:240-241 */
    int next = -1;
    float xyz[3];
    srlo_pose(c, s->H, s->O + (size_t)slot * c->object_res * c->object_res, u, v, xyz);   /* observer.py:400-403 */
    if (c->ordering_freedom) {               /* TestSimulator.step: pop the chosen rock (simulator.py:372-378) */
      int rk = slot / e->n_orient;
      s->pending = s->ids[rk];
      for (int k = rk; k + 1 < s->list_pos; ++k) s->ids[k] = s->ids[k + 1];
      s->list_pos -= 1;
      if (s->list_pos == 0) s->done = 1;     /* env.py:513-514: no objects left */
      else next = s->ids[0];
    } else if (s->list_pos < c->episode_length) next = s->ids[s->list_pos++];   /* env.py:243-247 */
    else s->done = 1;
    sim_step(e, s, V(xyz[0], xyz[1], xyz[2]), oi);
    s->pending = next;                                   /* _load, simulator.py:258 */
    render_heightmap(e, s->nb, s->mesh, s->x, s->q, s->H);
    observe_objects(e, s);
    step_rewards(e, s, reward + (size_t)i * K);
    done[i] = (uint8_t)s->done;
    pack_obs(e, s, obs_map + nm * i, obs_obj + no * i);
    if ((s->status & SRL_ST_DIVERGED) && rc == SRL_OK) rc = SRL_ESIM_DIVERGED;
  }
  if (rc == SRL_EINVAL_ACTION) fail(rc, "Invalid action.");
  if (rc == SRL_ESIM_DIVERGED) fail(rc, "Maximum number of simulator steps reached.");
  return rc;
}

int srlo_sample(srlo_env* e, int64_t* action) {
  const srl_config* c = &e->c;
  e->sample_counter += 1;
  for (int i = 0; i < c->n_envs; ++i) {
    uint32_t key = e->seed + (uint32_t)c->env_index_offset + (uint32_t)i;
    int nvalid = c->ordering_freedom ? (e->env[i].list_pos > 0 ? e->env[i].list_pos : 1) * e->n_orient : e->n_orient;
    action[i] = (int64_t)rng_below(srlo_rng(key, e->sample_counter, STREAM_ACTION, 0), (uint32_t)(e->A * nvalid));
  }
  return SRL_OK;
}

/* ------------------------------------------------------------------ telemetry */
int srlo_get_state(srlo_env* e, float* poses, int32_t* n_bodies, int32_t* substeps, int32_t* status) {
  for (int i = 0; i < e->c.n_envs; ++i) {
    env_t* s = &e->env[i];
    if (poses) {
      float* p = poses + (size_t)i * MAXB * 8;
      memset(p, 0, sizeof(float) * MAXB * 8);
      for (int b = 0; b < s->nb; ++b) {
        p[b * 8 + 0] = s->x[b].x; p[b * 8 + 1] = s->x[b].y; p[b * 8 + 2] = s->x[b].z;
        p[b * 8 + 3] = s->q[b].x; p[b * 8 + 4] = s->q[b].y; p[b * 8 + 5] = s->q[b].z; p[b * 8 + 6] = s->q[b].w;
        p[b * 8 + 7] = (float)s->mesh[b];
      }
    }
    if (n_bodies) n_bodies[i] = s->nb;
    if (substeps) { substeps[2 * i] = s->substeps[0]; substeps[2 * i + 1] = s->substeps[1]; }
    if (status) status[i] = s->status;
  }
  return SRL_OK;
}

int srlo_get_velocities(srlo_env* e, float* vel) {
  for (int i = 0; i < e->c.n_envs; ++i) {
    env_t* s = &e->env[i];
    float* p = vel + (size_t)i * MAXB * 8;
    memset(p, 0, sizeof(float) * MAXB * 8);
    for (int b = 0; b < s->nb; ++b) {
      p[b * 8 + 0] = s->v[b].x; p[b * 8 + 1] = s->v[b].y; p[b * 8 + 2] = s->v[b].z;
      p[b * 8 + 4] = s->w[b].x; p[b * 8 + 5] = s->w[b].y; p[b * 8 + 6] = s->w[b].z;
    }
  }
  return SRL_OK;
}

int srlo_get_contacts(srlo_env* e, float* max_pen, int32_t* n_points) {
  for (int i = 0; i < e->c.n_envs; ++i) {
    env_t* s = &e->env[i];
    float mp = 0.0f; int np = 0;
    for (int b = 0; b < s->nb; ++b)
      for (int k = 0; k < s->gm[b].np; ++k) { np++; if (-s->gm[b].dist[k] > mp) mp = -s->gm[b].dist[k]; }
    for (int sl = 0; sl < MAXSLOT; ++sl)
      if (s->pair_of_slot[sl] >= 0)
        for (int k = 0; k < s->man[sl].np; ++k) { np++; if (-s->man[sl].p[k].dist > mp) mp = -s->man[sl].p[k].dist; }
    if (max_pen) max_pen[i] = mp;
    if (n_points) n_points[i] = np;
  }
  return SRL_OK;
}

int srlo_get_sweeps(srlo_env* e, int32_t* sweeps) {
  for (int i = 0; i < e->c.n_envs; ++i) sweeps[i] = e->env[i].sweeps;
  return SRL_OK;
}

int srlo_get_maps(srlo_env* e, float* height, float* object_map, int32_t* goal_rect) {
  const srl_config* c = &e->c;
  size_t nh = (size_t)c->overhead_res * c->overhead_res, no = (size_t)c->object_res * c->object_res * e->n_slots;
  for (int i = 0; i < c->n_envs; ++i) {
    env_t* s = &e->env[i];
    if (height) memcpy(height + nh * i, s->H, nh * sizeof(float));
    if (object_map) memcpy(object_map + no * i, s->O, no * sizeof(float));
    if (goal_rect) for (int k = 0; k < 4; ++k) goal_rect[4 * i + k] = s->goal[k];
  }
  return SRL_OK;
}

int srlo_render_heightmap(srlo_env* e, const float* poses, const int32_t* mesh_ids, int32_t nb, float* height) {
  if (!e->mesh) return fail(SRL_ENOMESH, "load meshes first");
  if (nb < 0 || nb > MAXB) return fail(SRL_EINVAL, "n_bodies out of range");
  int mesh[MAXB]; v3 x[MAXB]; q4 q[MAXB];
  for (int b = 0; b < nb; ++b) {
    if (mesh_ids[b] < 0 || mesh_ids[b] >= e->n_mesh) return fail(SRL_EINVAL, "mesh id out of range");
    mesh[b] = mesh_ids[b];
    x[b] = V(poses[7 * b], poses[7 * b + 1], poses[7 * b + 2]);
    q[b].x = poses[7 * b + 3]; q[b].y = poses[7 * b + 4]; q[b].z = poses[7 * b + 5]; q[b].w = poses[7 * b + 6];
  }
  render_heightmap(e, nb, mesh, x, q, height);
  return SRL_OK;
}

/* The plain statement of O1 (render_heightmap_all) and, for the comparison before the codec, the culled definition's raw
 * heights.  form: 0 = the culled definition (item rows, ranges, spans), 1 = all faces / all outline sides / every pixel of
 * the bounding box, 2 = z_lo <= z_hi over all planes.  raw != 0: heights before the depth codec. */
int srlo_render_heightmap_all(srlo_env* e, const float* poses, const int32_t* mesh_ids, int32_t nb, int32_t form,
                              int32_t raw, float* height) {
  if (!e->mesh) return fail(SRL_ENOMESH, "load meshes first");
  if (nb < 0 || nb > MAXB) return fail(SRL_EINVAL, "n_bodies out of range");
  if (form < 0 || form > 2) return fail(SRL_EINVAL, "form is 0 (culled), 1 (plain outline) or 2 (hull interval)");
  int mesh[MAXB]; v3 x[MAXB]; q4 q[MAXB];
  for (int b = 0; b < nb; ++b) {
    if (mesh_ids[b] < 0 || mesh_ids[b] >= e->n_mesh) return fail(SRL_EINVAL, "mesh id out of range");
    mesh[b] = mesh_ids[b];
    x[b] = V(poses[7 * b], poses[7 * b + 1], poses[7 * b + 2]);
    q[b].x = poses[7 * b + 3]; q[b].y = poses[7 * b + 4]; q[b].z = poses[7 * b + 5]; q[b].w = poses[7 * b + 6];
  }
  if (form == 0) { render_heightmap_raw(e, nb, mesh, x, q, height); if (!raw) heightmap_codec(e, height); }
  else render_heightmap_all(e, nb, mesh, x, q, form, raw, height);
  return SRL_OK;
}

int srlo_render_object(srlo_env* e, int32_t mesh_id, float* object_map) {
  if (!e->mesh) return fail(SRL_ENOMESH, "load meshes first");
  if (mesh_id >= e->n_mesh) return fail(SRL_EINVAL, "mesh id out of range");
  render_object(e, mesh_id, object_map);
  return SRL_OK;
}

/* debug / invariant hook: run `n` raw sub-steps on env `i` (no stop criterion) */
int srlo_set_body_state(srlo_env* e, const float* poses, const float* vel) {
  for (int i = 0; i < e->c.n_envs; ++i) {
    env_t* s = &e->env[i];
    for (int b = 0; b < s->nb; ++b) {
      if (poses) {
        const float* p = poses + ((size_t)i * MAXB + b) * 8;
        s->x[b] = V(p[0], p[1], p[2]);
        s->q[b].x = p[3]; s->q[b].y = p[4]; s->q[b].z = p[5]; s->q[b].w = p[6];
      }
      if (vel) {
        const float* p = vel + ((size_t)i * MAXB + b) * 8;
        s->v[b] = V(p[0], p[1], p[2]); s->w[b] = V(p[4], p[5], p[6]);
      }
    }
  }
  return SRL_OK;
}

int srlo_step_simulation(srlo_env* e, int32_t n) {
  for (int i = 0; i < e->c.n_envs; ++i) {
    env_t* s = &e->env[i];
    if (s->nb == 0) continue;
    s->sweeps = 0;
    for (int k = 0; k < n; ++k) substep(e, s);
  }
  return SRL_OK;
}

int srlo_debug_substeps(srlo_env* e, int32_t i, int32_t n) {
  if (i < 0 || i >= e->c.n_envs) return fail(SRL_EINVAL, "env index");
  for (int k = 0; k < n; ++k) substep(e, &e->env[i]);
  return SRL_OK;
}

int srlo_debug_dump(srlo_env* e, int32_t i) {
  env_t* s = &e->env[i];
  for (int b = 0; b < s->nb; ++b) {
    printf(" body %d mesh %d x=(%.5f %.5f %.5f) v=(%.5f %.5f %.5f) w=(%.4f %.4f %.4f)\n", b, s->mesh[b], s->x[b].x, s->x[b].y, s->x[b].z,
           s->v[b].x, s->v[b].y, s->v[b].z, s->w[b].x, s->w[b].y, s->w[b].z);
    for (int k = 0; k < s->gm[b].np; ++k)
      printf("   ground vid %d dist %.6f in %.6f it %.6f %.6f\n", s->gm[b].vid[k], s->gm[b].dist[k], s->gm[b].in[k], s->gm[b].it1[k], s->gm[b].it2[k]);
  }
  for (int sl = 0; sl < MAXSLOT; ++sl) if (s->pair_of_slot[sl] >= 0) {
    manifold_t* m = &s->man[sl];
    printf(" slot %d (%d,%d) colour %d np %d\n", sl, s->slot_a[sl], s->slot_b[sl], s->colour[sl], m->np);
    for (int k = 0; k < m->np; ++k)
      printf("   dist %.6f n=(%.4f %.4f %.4f) la=(%.4f %.4f %.4f) in %.6f it %.6f %.6f\n", m->p[k].dist, m->p[k].n.x, m->p[k].n.y, m->p[k].n.z,
             m->p[k].la.x, m->p[k].la.y, m->p[k].la.z, m->p[k].in, m->p[k].it1, m->p[k].it2);
  }
  fflush(stdout);
  return 0;
}

/* debug: slot statistics of env i: active slots, slots with contact points, total points, colours */
int srlo_debug_slots(srlo_env* e, int32_t i, int32_t* out4) {
  env_t* s = &e->env[i];
  int act = 0, con = 0, pts = 0;
  for (int sl = 0; sl < MAXSLOT; ++sl) if (s->pair_of_slot[sl] >= 0) { act++; if (s->man[sl].np > 0) con++; pts += s->man[sl].np; }
  out4[0] = act; out4[1] = con; out4[2] = pts; out4[3] = s->ncolour;
  return 0;
}
