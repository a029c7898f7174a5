// stage.h — what the ray cast of srl_k_render needs of ONE rock, prepared by one wave: shared by srl_k_stage (render.hip: one
// wave per (env, rock), the explicit-pose hook and the fall-back) and by the tail of the settle kernels (settle.hip), where an
// env's workgroup stages its rocks as soon as its stop criterion has fired — all but the launch's slowest env finish long
// before the launch ends and their SIMDs idle, so the staging costs the step only what the LAST env's rocks take (round 5;
// until then srl_k_stage ran between srl_k_step and srl_k_render, 13.3 us per step).  One statement of the arithmetic
// (stage_compute), so the records are the same bits wherever they are made.
#pragma once
#include "srl_device.h"
#include "srl_kernels.h"

// world-frame render plane of one face: z = a x + b y + c; w = 0 up-facing (z_hi = min), 1 down-facing
// (z_lo = max).  |n_z| is clamped to >= 1e-6: a vertical face becomes a plane of enormous slope that never
// limits z on its inner side and empties the interval on its outer side.
#if SRL_BISECT & 4
SRL_HELPER(2) float4 make_rplane(float4 pl, const m3& R, v3 x) {
#else
__device__ __forceinline__ float4 make_rplane(float4 pl, const m3& R, v3 x) {
#endif
  v3 nw = mmul(R, V(pl.x, pl.y, pl.z));
  float dw = pl.w + dot(nw, x);
  float nz = nw.z;
  int type;
  if (nz >= 0.0f) { if (nz < 1e-6f) nz = 1e-6f; type = 0; }
  else { if (nz > -1e-6f) nz = -1e-6f; type = 1; }
  return make_float4(-nw.x / nz, -nw.y / nz, dw / nz, __int_as_float(type));
}

__device__ __forceinline__ bool pixel_range(float lo, float hi, float inv_px, int res, int& i0, int& i1) {
  float f0 = ceilf(lo * inv_px - 0.5f), f1 = floorf(hi * inv_px - 0.5f);
  if (f0 < 0.0f) f0 = 0.0f;
  if (f1 > (float)(res - 1)) f1 = (float)(res - 1);
  if (f1 < f0) return false;
  i0 = (int)f0; i1 = (int)f1;
  return true;
}

#ifndef SRL_ITEM_ROWS
#define SRL_ITEM_ROWS 4   // a lane's item: SRL_ITEM_ROWS rows x 2 columns of pixels (one plane fetch serves them all)
#endif

// ---- the staged record of one rock (srl_k_stage -> srl_k_render), SRL_STAGE_STRIDE float4s per (env, body slot) in HBM:
//   [0] ints  i0 | i1 << 16, j0 | j1 << 16 (pixel bounding box), nup (up-facing planes), nsil (outline sides)
//   [1] ints  items (of SRL_ITEM_ROWS x 2 pixels), nir (item rows with an entry below; 0: items fill the bounding box and
//             every item sweeps the whole lists), 0, 0
//   [2..5]    16 ints, one per item row r: first item of the row << 8 | first column — the columns of the bounding box
//             the outline can reach in the row's SRL_ITEM_ROWS pixel rows (a superset: the ray cast still tests every pixel)
//   [6..9]    16 ints, one per item row: the ranges of the two lists an item of the row sweeps, as bytes
//             plane_lo | plane_hi << 8 | side_lo << 16 | side_hi << 24 (a superset of the faces / outline edges whose
//             x extent reaches the row, see srl_k_stage)
//   [10..]    the nup up-facing world-frame planes (a, b, c, -), then the nsil outline sides (ea, eb, ec, -), each list
//             ordered by the first item row its face / edge reaches
#define SRL_STAGE_HDR 10
#define SRL_STAGE_SPANS 16
#define SRL_STAGE_STRIDE (SRL_STAGE_HDR + SRL_MAX_TRIS + 2)

// min / max over the 64 lanes of a wave by DPP row shifts and row broadcasts (no LDS traffic); the result is
// returned to every lane through an SGPR.  min / max are idempotent, so lanes without a source keep their own value.
template <bool MIN>
__device__ __forceinline__ float wave_minmax(float v) {
#define SRL_DPP_STEP(ctrl, rows)                                                                                      \
  {                                                                                                                   \
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rows, 0xf, false)); \
    v = MIN ? fminf(v, o) : fmaxf(v, o);                                                                              \
  }
  SRL_DPP_STEP(0x111, 0xf) SRL_DPP_STEP(0x112, 0xf) SRL_DPP_STEP(0x114, 0xf) SRL_DPP_STEP(0x118, 0xf)   // row_shr:1,2,4,8
  SRL_DPP_STEP(0x142, 0xa) SRL_DPP_STEP(0x143, 0xc)                                                      // row_bcast:15, :31
#undef SRL_DPP_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}


// ------------------------------------------------------------------ staging: one wave per (env, rock)
// What the ray cast needs of a rock — its pixel bounding box, its up-facing face planes in the world frame, the sides of
// its outline, and per item row the columns the outline can reach and the faces / sides that matter there — depends on the
// rock's pose and mesh alone.  Until round 4 srl_k_render did this itself, two wave tasks per rock between two block
// barriers, with a 64 KB tile per workgroup limiting a CU to 16 waves: four dependent memory round trips that two
// workgroups per CU could not hide (7 us of a 43 us launch).  Here the same arithmetic runs with one wave per rock and no
// tile, and srl_k_render starts from finished records (one contiguous read per rock).
//   xy bounds: vertices over the lanes, DPP min / max -> pixel range (pixel_range)
//   planes:    world-frame plane of every face (make_rplane), the up-facing ones kept
//   outline:   lanes over the mesh's edge list (edge -> its two faces, built at srl_load_meshes); an edge between an up- and
//              a down-facing face (facing = the bit make_rplane classifies by, looked up in the ballot masks of the face
//              pass) is a side of the outline: through the projected end points A, B (A the lower vertex index)
//              E(p) = fma(ea, p.x, fma(eb, p.y, ec)) >= 0 inside, oriented by the centre of mass
//   spans:     lane r = item row r (SRL_ITEM_ROWS pixel rows, x in [xa, xb]): a side with eb > 0 bounds y from below by
//              its line, one with eb < 0 from above; over the slab a line is at least / at most its value at one of the
//              two ends, so max_k min(L_k(xa), L_k(xb)) <= y <= min_k max(L_k(xa), L_k(xb)) holds for every inside
//              point (widened by 1e-4 m against rounding).  A superset is all that is needed: the ray cast evaluates the
//              definition's side functions at every pixel it visits.
//   ranges:    along the vertical line through a pixel inside the outline the hull's top is the face the line pierces, so
//              min over ALL up-facing planes = min over any subset that holds that face: an item row needs the faces whose
//              x extent reaches its rows (rounded outwards to whole pixel rows: a neighbour that a rounding could prefer at
//              the face's rim is included).  The
//              same holds for the outline: a point of the row outside it violates a side whose x extent contains the
//              point's x, or one of the two sides at the outline's extreme vertex.  Both lists are ordered by the first
//              item row their face / edge reaches (LDS counters), so a row's subset is one range [lo, hi) of each.
// Lists are sets here: the ray cast takes minima over them, so their order does not enter any result.
struct StageLds {
  float4 slot[SRL_MAX_TRIS + 2];     // the rock's up-facing planes from the front, its outline sides from the back; .w = r0 | r1 << 8
  int cnt[2][SRL_STAGE_SPANS], st[2][SRL_STAGE_SPANS], lo[2][SRL_STAGE_SPANS];   // entries per first row, their prefix, first row of the entries that reach a row
  float2 wxy[SRL_MAX_VERTS];         // world xy of the vertices: faces and edges read their end points by vertex id
};
static_assert(SRL_MAX_VERTS == 128, "vertex ids are masked with 127 below");

// the scalars of DevParams the staging reads (a by-value DevParams in srl_k_stage, a pointer to it in the settle kernels)
struct StageArgs {
  const float4* mp; const uchar4* mt; const uchar4* me; const float4* mv;
  float px, inv_px; int res;
};

// item rows [r0, r1] (clamped to the rock's) that a face / edge with x extent [x0, x1] reaches: those of the pixel rows from
// the last one whose centre is at or below x0 to the first one at or above x1 (so up to a pixel of slack on either side)
__device__ __forceinline__ int slab_range(float x0, float x1, float inv_px, int i0, int nirows) {
  const float f0 = floorf(x0 * inv_px - 0.5f) - (float)i0, f1 = ceilf(x1 * inv_px - 0.5f) - (float)i0;
  int r0 = f0 > 0.0f ? (int)(f0 * (1.0f / SRL_ITEM_ROWS)) : 0, r1 = f1 > 0.0f ? (int)(f1 * (1.0f / SRL_ITEM_ROWS)) : 0;
  r0 = min(r0, nirows - 1); r1 = min(r1, nirows - 1);
  return r0 | (r1 << 8);
}

// Everything the mesh header points at, requested in one go (one memory round trip): the first two chunks of faces (planes +
// vertex ids), the first three chunks of edges, the vertices (two per lane).  A caller with several rocks to stage requests
// the next rock's while it computes the current one's (settle.hip).
struct StageLoads {
  float4 pl0, pl1, lv0, lv1;
  uchar4 tr0, tr1, ed0, ed1, ed2;
};
__device__ __forceinline__ StageLoads stage_request(const StageArgs& A, const MeshHdr& mh, int lane) {
  StageLoads g;
  g.pl0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); g.pl1 = g.pl0; g.lv0 = g.pl0; g.lv1 = g.pl0;
  g.tr0 = make_uchar4(0, 0, 0, 0); g.tr1 = g.tr0; g.ed0 = g.tr0; g.ed1 = g.tr0; g.ed2 = g.tr0;
  const int vo = mh.vo, nv = mh.nv, to = mh.to, nt = mh.nt, eo = mh.eo, ne = mh.ne;
  if (lane < nt) { g.pl0 = A.mp[to + lane]; g.tr0 = A.mt[to + lane]; }
  if (64 + lane < nt) { g.pl1 = A.mp[to + 64 + lane]; g.tr1 = A.mt[to + 64 + lane]; }
  if (lane < ne) g.ed0 = A.me[eo + lane];
  if (64 + lane < ne) g.ed1 = A.me[eo + 64 + lane];
  if (128 + lane < ne) g.ed2 = A.me[eo + 128 + lane];
  if (lane < nv) g.lv0 = A.mv[vo + lane];
  if (64 + lane < nv) g.lv1 = A.mv[vo + 64 + lane];
  return g;
}

// One rock: pose (xb, q), mesh header, the requested geometry -> its record `rec` (SRL_STAGE_STRIDE float4s).  One wave; S is
// the wave's own LDS area; no block-wide synchronisation.
__device__ __forceinline__ void stage_compute(const StageArgs& A, StageLds& S, float4* __restrict__ rec, v3 xb, q4 q,
                                              const MeshHdr& mh, const StageLoads& g, int lane) {
  const int nv = mh.nv, to = mh.to, nt = mh.nt, eo = mh.eo, ne = mh.ne;
  const int res = A.res;
  const m3 R = quat_to_mat(q);
  // world xy of the vertices into LDS, one (two) per lane; faces and edges read their end points there by vertex id (one
  // 8-byte LDS read per end point; as cross-lane reads of per-lane copies they were up to eight ds_bpermute per chunk)
  float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    const int v = 64 * ch + lane;
    if (v < nv) {
      const float4 lv = ch == 0 ? g.lv0 : g.lv1;
      const v3 a = mmul_add(R, V(lv.x, lv.y, lv.z), xb);
      S.wxy[v] = make_float2(a.x, a.y);
      xmin = fminf(xmin, a.x); xmax = fmaxf(xmax, a.x); ymin = fminf(ymin, a.y); ymax = fmaxf(ymax, a.y);
    }
  }
  __builtin_amdgcn_wave_barrier();                   // (one wave: LDS keeps its order; the compiler must too)
  xmin = wave_minmax<true>(xmin); xmax = wave_minmax<false>(xmax);
  ymin = wave_minmax<true>(ymin); ymax = wave_minmax<false>(ymax);
  int i0 = 0, i1 = -1, j0 = 0, j1 = -1;
  const bool okx = pixel_range(xmin, xmax, A.inv_px, res, i0, i1);
  const bool oky = pixel_range(ymin, ymax, A.inv_px, res, j0, j1);
  if (!(okx && oky)) {                               // the rock lies outside the window: nothing to cast
    if (lane == 0) {
      rec[0] = make_float4(__int_as_float(-65536), __int_as_float(-65536), __int_as_float(0), __int_as_float(0));   // i0 = 0, i1 = -1
      rec[1] = make_float4(__int_as_float(0), __int_as_float(0), __int_as_float(0), __int_as_float(0));
    }
    return;
  }
  const int nirows = (i1 - i0 + SRL_ITEM_ROWS) / SRL_ITEM_ROWS;
  const int nir = nirows <= SRL_STAGE_SPANS ? nirows : 0;   // more item rows than the tables hold: one row for the lists
  const unsigned long long below = (1ull << lane) - 1ull;
  if (lane < SRL_STAGE_SPANS) { S.cnt[0][lane] = 0; S.cnt[1][lane] = 0; S.lo[0][lane] = 0x7fffffff; S.lo[1][lane] = 0x7fffffff; }
  // ---- up-facing planes -> S.slot[0 .. nup), their facing kept as ballot masks for the outline pass
  unsigned long long upm0 = 0ull, upm1 = 0ull, upm2 = 0ull, upm3 = 0ull;
  int nup = 0;
#pragma unroll
  for (int ch = 0; ch < (SRL_MAX_TRIS + 63) / 64; ++ch) {
    const int c = 64 * ch;
    if (c < nt) {
      const bool act = c + lane < nt;
      float4 pl = ch == 0 ? g.pl0 : g.pl1;
      uchar4 tr = ch == 0 ? g.tr0 : g.tr1;
      if (ch >= 2 && act) { pl = A.mp[to + c + lane]; tr = A.mt[to + c + lane]; }
      const float4 wp = make_rplane(pl, R, xb);
      const bool up = act && __float_as_int(wp.w) == 0;
      const unsigned long long mu = __ballot(up);
      if (ch == 0) upm0 = mu; else if (ch == 1) upm1 = mu; else if (ch == 2) upm2 = mu; else upm3 = mu;
      // world x of the face's three vertices
      const float fa = S.wxy[tr.x & 127].x, fb = S.wxy[tr.y & 127].x, fc = S.wxy[tr.z & 127].x;
      if (up) {
        const int rr = nir ? slab_range(fminf(fa, fminf(fb, fc)), fmaxf(fa, fmaxf(fb, fc)), A.inv_px, i0, nirows) : 0;
        S.slot[nup + __popcll(mu & below)] = make_float4(wp.x, wp.y, wp.z, __int_as_float(rr));
        atomicAdd(&S.cnt[0][rr & 0xff], 1);
        atomicMin(&S.lo[0][rr >> 8], rr & 0xff);
      }
      nup += __popcll(mu);
    }
  }
  // ---- outline sides -> S.slot[nt + 1 - k] (from the back; at most nt + 2 - nup: a closed triangulated cap with an
  //      s-edge rim has >= s - 2 triangles)
  const int cap = nt + 2 - nup;
  int nsil = 0;
  for (int c = 0; c < ne; c += 64) {
    const bool act = c + lane < ne;
    uchar4 ed = c == 0 ? g.ed0 : c == 64 ? g.ed1 : g.ed2;
    if (c >= 192) { ed = make_uchar4(0, 0, 0, 0); if (act) ed = A.me[eo + c + lane]; }
    const unsigned long long ma = ed.z < 128 ? (ed.z < 64 ? upm0 : upm1) : (ed.z < 192 ? upm2 : upm3);
    const unsigned long long mb = ed.w < 128 ? (ed.w < 64 ? upm0 : upm1) : (ed.w < 192 ? upm2 : upm3);
    const bool ua = (ma >> (ed.z & 63)) & 1ull, ub = (mb >> (ed.w & 63)) & 1ull;
    const bool sil = act && (ua != ub);
    const unsigned long long ms = __ballot(sil);
    const float2 pa = S.wxy[ed.x & 127], pbv = S.wxy[ed.y & 127];
    const float Ax = pa.x, Ay = pa.y, Bx = pbv.x, By = pbv.y;
    const int sidx = nsil + __popcll(ms & below);
    if (sil && sidx < cap) {
      float ea = Ay - By, eb = Bx - Ax;
      float ec = -fmaf(ea, Ax, eb * Ay);
      if (fmaf(ea, xb.x, fmaf(eb, xb.y, ec)) < 0.0f) { ea = -ea; eb = -eb; ec = -ec; }
      const int rr = nir ? slab_range(fminf(Ax, Bx), fmaxf(Ax, Bx), A.inv_px, i0, nirows) : 0;
      S.slot[nt + 1 - sidx] = make_float4(ea, eb, ec, __int_as_float(rr));
      atomicAdd(&S.cnt[1][rr & 0xff], 1);
      atomicMin(&S.lo[1][rr >> 8], rr & 0xff);
    }
    nsil += __popcll(ms);
  }
  if (nsil > cap) nsil = cap;
  __builtin_amdgcn_wave_barrier();                   // (one wave: LDS keeps its order; the compiler must too)
  // ---- items: SRL_ITEM_ROWS x 2 pixels; per item row the columns the outline can reach
  int items;
  if (nir) {
    // lane = 4 (item row) + (quarter of the sides); the quarters combine by DPP within the quad (max / min / or)
    const int row = lane >> 2, sq = lane & 3;
    const int ia = i0 + SRL_ITEM_ROWS * row, ib = min(ia + SRL_ITEM_ROWS - 1, i1);
    const float xa = ((float)ia + 0.5f) * A.px, xe = ((float)ib + 0.5f) * A.px;
    float ylo = -1e30f, yhi = 1e30f;
    int empty = 0;
    for (int k = sq; k < nsil; k += 4) {
      const float4 sd = S.slot[nt + 1 - k];
      const float fa = fmaf(sd.x, xa, sd.z), fe = fmaf(sd.x, xe, sd.z);   // E = f + eb y
      if (fabsf(sd.y) < 1e-12f) { empty |= (fa < 0.0f && fe < 0.0f) ? 1 : 0; continue; }
      const float rcp = -1.0f / sd.y;                                        // (IEEE division: the oracle states the same bounds)
      const float la = fa * rcp, le = fe * rcp;                              // the side's line at the two ends of the slab
      if (sd.y > 0.0f) ylo = fmaxf(ylo, fminf(la, le)); else yhi = fminf(yhi, fmaxf(la, le));
    }
#define SRL_QUAD(ctrl)                                                                                                     \
    ylo = fmaxf(ylo, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ylo), __float_as_int(ylo), ctrl, 0xf, 0xf, false))); \
    yhi = fminf(yhi, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(yhi), __float_as_int(yhi), ctrl, 0xf, 0xf, false))); \
    empty |= __builtin_amdgcn_update_dpp(empty, empty, ctrl, 0xf, 0xf, false);
    SRL_QUAD(0xB1) SRL_QUAD(0x4E)                    // quad_perm [1,0,3,2], [2,3,0,1]
#undef SRL_QUAD
    int cnt = 0, jlo = j0;
    if (row < nirows) {
      // pixel centres ((j + 0.5) px) between the bounds widened by 1e-4 m, inside the bounding box
      const float fl = ceilf((ylo - 1e-4f) * A.inv_px - 0.5f), fh = floorf((yhi + 1e-4f) * A.inv_px - 0.5f);
      const int a = fl > (float)j0 ? (fl < (float)(j1 + 1) ? (int)fl : j1 + 1) : j0;
      const int z = fh < (float)j1 ? (fh > (float)(j0 - 1) ? (int)fh : j0 - 1) : j1;
      if (!empty && a <= z) { jlo = a; cnt = (z - a + 2) >> 1; }
    }
    int pre = cnt;                                   // inclusive prefix over the item rows (every lane of a row's quad holds it)
#pragma unroll
    for (int d = 4; d < 64; d <<= 1) { const int v = __shfl_up(pre, d); if (lane >= d) pre += v; }
    items = __shfl(pre, 63);
    if (sq == 0) ((int*)(rec + 2))[row] = ((pre - cnt) << 8) | jlo;
  } else {                                           // items fill the bounding box
    items = nirows * ((j1 - j0 + 2) >> 1);
  }
  // ---- the two lists into the record, ordered by first item row (within one first row in no particular order); an item row
  //      r consults the entries whose first row lies in [R(r), r], R(r) = the smallest first row among the entries that reach
  //      r — a set that does not depend on the order inside the buckets, stated the same way by the oracle
  int hi0 = 0, hi1 = 0, lo0 = 0, lo1 = 0;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    int c = 0, incl = 0, rmin = 0x7fffffff;
    if (lane < SRL_STAGE_SPANS) {
      c = S.cnt[w][lane]; incl = c; rmin = S.lo[w][lane];
      // inclusive prefix of the counts and suffix minimum of the first rows over the 16 lanes = one DPP row: shifts by 1, 2, 4,
      // 8 with zero / INT_MAX shifted in (no ds_bpermute)
      static_assert(SRL_STAGE_SPANS == 16, "the scans below run over one DPP row");
#define SRL_SCAN(ctrl_r, ctrl_l)                                                                 \
      incl += __builtin_amdgcn_update_dpp(0, incl, ctrl_r, 0xf, 0xf, true);                      \
      rmin = min(rmin, __builtin_amdgcn_update_dpp(0x7fffffff, rmin, ctrl_l, 0xf, 0xf, false));
      SRL_SCAN(0x111, 0x101) SRL_SCAN(0x112, 0x102) SRL_SCAN(0x114, 0x104) SRL_SCAN(0x118, 0x108)
#undef SRL_SCAN
      S.st[w][lane] = incl - c; S.cnt[w][lane] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    int lo = incl;                                   // (no entry reaches the row: an empty range)
    if (lane < SRL_STAGE_SPANS && rmin <= lane) lo = S.st[w][rmin];
    if (w == 0) { hi0 = incl; lo0 = lo; } else { hi1 = incl; lo1 = lo; }
    const int n = w == 0 ? nup : nsil;
    for (int k = lane; k < n; k += 64) {
      const float4 v = S.slot[w == 0 ? k : nt + 1 - k];
      const int r0 = __float_as_int(v.w) & 0xff;
      const int pos = S.st[w][r0] + atomicAdd(&S.cnt[w][r0], 1);
      rec[SRL_STAGE_HDR + (w == 0 ? 0 : nup) + pos] = make_float4(v.x, v.y, v.z, 0.0f);
    }
  }
  if (lane < SRL_STAGE_SPANS) ((int*)(rec + 6))[lane] = lo0 | (hi0 << 8) | (lo1 << 16) | (hi1 << 24);
  if (lane == 0) {
    if (items == 0) { nup = 0; nsil = 0; }
    rec[0] = make_float4(__int_as_float(i0 | (i1 << 16)), __int_as_float(j0 | (j1 << 16)), __int_as_float(nup), __int_as_float(nsil));
    rec[1] = make_float4(__int_as_float(items), __int_as_float(nir), __int_as_float(0), __int_as_float(0));
  }
}
