// render.hip — K2 (overhead height map) + K5 (reward) + observation packing, and K3 (object map).
//
// Renderer definition: convex ray cast.  A rock is the intersection of its face
// half-spaces; along the vertical line through a pixel centre the hull spans [z_lo, z_hi] with
// z_hi = min over up-facing planes and z_lo = max over down-facing planes.  The overhead camera
// (observer.py:252-260) sees z_hi inside the rock's outline (the projected mesh edges between an up- and a
// down-facing face; same set as z_lo <= z_hi), the object camera from below (observer.py:262-277) sees z_lo.
//
// K2: one 512-thread workgroup per env.  The res x res tile lives in LDS (64 KB at 128^2 -> 2 workgroups = 16
// waves per CU).  Prologue (speculative pose / mesh-header loads, tile zeroing) -> two wave tasks per rock: xy
// bounds by DPP min / max + the up-facing planes packed from the front of the rock's LDS region, and the outline
// sides packed from its back (ballot ranks, no atomics) -> rows no rock reaches are written out
// at once -> ray cast: lanes over the flattened (rock, item of 4 x 2 pixels) list sweep the rock's planes (LDS
// broadcasts, packed FMAs, min3) and merge into the tile — as rows of the depth-codec table, which are monotone in the
// height — with integer atomicMax -> one branch-free epilogue pass looks the reference's depth codec up
// (observer.py:259-260, tabulated over the float32 lattice of 1000 - z: DevParams::codec), streams out
// H (16 B per lane, 1 KB contiguous per wave store), the packed uint8 observation (env.py:171-172, :228-231) and
// accumulates the IoU sums (rewarder.py:297-307) in the fixed order DESIGN.md defines.  HBM traffic per env step
// is the algorithmic 6*res^2 + 5*r^2 bytes out plus ~1.4 KB per rock of mesh/pose data in (L2-resident pool).
//
// K3: per-mesh underside map rendered once at srl_load_meshes and cached: it depends only on the mesh
// (spawn orientation is the identity, env.py:120-121).
#include "srl_device.h"
#include "srl_kernels.h"
#include "stage.h"

#define SRL_RENDER_THREADS 512

// Diagnostic builds only (tools/stamps_render.py, tools/ab_render.py; the outputs of the ablations are wrong by design):
//   SRL_STAMPS        per-phase wall-clock stamps of thread 0
//   SRL_ABL_NOSTORE   no H / observation stores     SRL_ABL_NOCAST  no ray cast     SRL_ABL_NOSTAGE  no staging, no ray cast
//   SRL_ABL_EMPTY     the kernel returns at once (launch + event floor)
#ifdef SRL_STAMPS
#define RSTAMP(k) do { if (tid == 0) { long long _t = wall_clock64(); P.hdr[e].rstamps[k] += _t - _t0; _t0 = _t; } } while (0)
#else
#define RSTAMP(k)
#endif

__device__ __forceinline__ float elev_overhead(const DevParams& P, float d) {
  return SRL_FAR - P.elev_num / (SRL_FAR - P.c.max_z * d);
}
__device__ __forceinline__ float elev_object(const DevParams& P, float d) {
  return P.obj_c1 - P.obj_c2 / (SRL_FAR + P.c.object_max_dimension * (0.5f - d));
}

// the overhead depth codec tabulated over the lattice of t = fl(FAR - z) (see DevParams::codec): row 0 holds the
// elevation bits and the observation byte of a pixel that saw no rock, row v in 1 .. n those for t = near + (n - v) 2^-14
extern "C" __global__ void __launch_bounds__(256) srl_k_codec_table(DevParams P, uint2* __restrict__ tab, int n) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  const float nearp = SRL_FAR - P.c.max_z;
  const float den = fmaxf(P.c.max_z, P.c.object_max_dimension);
  if (v == n + 1) {   // goal-channel bytes (env.py:171-172 applied to the goal height and to 0)
    tab[v] = make_uint2((uint32_t)(uint8_t)((P.goal_z * 255.0f) / den), (uint32_t)(uint8_t)((0.0f * 255.0f) / den));
    return;
  }
  if (v == n + 2) {   // byte of an empty object-map pixel
    tab[v] = make_uint2((uint32_t)(uint8_t)((elev_object(P, 1.0f) * 255.0f) / den), 0u);
    return;
  }
  if (v > n) return;
  const float t = v > 0 ? nearp + (float)(n - v) * (1.0f / 16384.0f) : SRL_FAR - 0.0f;   // exact: a lattice point of [512, 1024]
  const float hh = elev_overhead(P, depth_encode(t, nearp, SRL_FAR));
  tab[v] = make_uint2(__float_as_uint(hh), (uint32_t)(uint8_t)((hh * 255.0f) / den));
}

// reference evaluation over an unsorted plane list (K3); K2 uses the type-sorted loops below.
// The fused multiply-adds are part of the definition (the oracle calls fmaf()).
__device__ __forceinline__ bool ray_cast(const float4* pl, int n, float px, float py, float& lo, float& hi) {
  hi = 1e30f; lo = -1e30f;
  for (int t = 0; t < n; ++t) {
    float4 p = pl[t];
    float z = fmaf(p.x, px, fmaf(p.y, py, p.z));
    if (__float_as_int(p.w) == 0) hi = fminf(hi, z);
    else lo = fmaxf(lo, z);
  }
  return lo <= hi;
}

// discount of one body (rewarder.py:261-269), thread 0 only
__device__ float body_discount(const DevParams& P, const float* gb, int b) {
  const float pmax = (float)P.c.object_res * P.px;   // rewarder.py:126
  const float omax = 3.14159265358979f;
  const v3 dp = ld3(gb + P.OFF_PX + 4 * b) - ld3(gb + P.OFF_X + 4 * b);
  const float perr = sqrtf(dot(dp, dp));
  const float* a = gb + P.OFF_PQ + 4 * b;
  const float* q = gb + P.OFF_Q + 4 * b;
  const float dw = fabsf((a[0] * q[0] + a[1] * q[1]) + (a[2] * q[2] + a[3] * q[3]));
  const float oerr = 2.0f * srl_acosf(fminf(dw, 1.0f));
  float disc = 1.0f;
  if (P.c.reward_pexp >= 0) {
    float t = perr / pmax, pw = 1.0f;
    for (int k = 0; k < P.c.reward_pexp; ++k) pw = pw * t;
    disc = disc * fmaxf(0.0f, 1.0f - pw);
  }
  if (P.c.reward_oexp >= 0) {
    float t = oerr / omax, pw = 1.0f;
    for (int k = 0; k < P.c.reward_oexp; ++k) pw = pw * t;
    disc = disc * fmaxf(0.0f, 1.0f - pw);
  }
  return disc;
}

// Rewarder.call (rewarder.py:162-179) for the discounted metrics (DOR or DIoU), thread 0 only
__device__ float discounted_metric(const DevParams& P, const EnvHdr* h, const float* gb, int metric) {
  float r = 0.0f; int nout = 0;
  for (int b = 0; b < h->nb; ++b) {
    v3 x = ld3(gb + P.OFF_X + 4 * b);
    float fu = floorf(x.x / P.px), fv = floorf(x.y / P.px);   // xy_to_pixel (observer.py:388-390)
    bool in = fu >= (float)h->goal[0] && fv >= (float)h->goal[1] && fu < (float)(h->goal[0] + h->goal[2]) &&
              fv < (float)(h->goal[1] + h->goal[3]);
    if (!in) { nout++; continue; }
    r = r + body_discount(P, gb, b);
  }
  if (metric == SRL_METRIC_DOR) return r / (float)P.c.episode_length;
  return r / (float)(P.c.episode_length + nout);
}

// average discount of all rocks (`_discounted(intersection=False)`, rewarder.py:149-151), thread 0 only
__device__ float average_discount(const DevParams& P, const EnvHdr* h, const float* gb) {
  float d = 0.0f;
  for (int b = 0; b < h->nb; ++b) d = d + body_discount(P, gb, b);
  return d / (float)h->nb;
}

// row of the codec table (DevParams::codec) for a ray-cast height z > 0: n - k with k the index of t = fl(FAR - z) on the
// float32 lattice above near (clamped to near: a rock above the window), so that higher z = higher row
__device__ __forceinline__ int codec_row(float z, float nearp, int n) {
  const float tt = fmaxf(SRL_FAR - z, nearp);
  return n - (int)((tt - nearp) * 16384.0f);
}

// which of the four pixels (row i, columns jb .. jb + 3) lie in the goal rectangle rows [g0, g0 + g2) x columns
// [g1, g1 + g3): bit t = pixel jb + t.  Branch-free: the epilogue is bound by VALU issue, not by memory.
__device__ __forceinline__ uint32_t goal_mask4(int i, int jb, int g0, int g2, int g1, int g3) {
  const int lo = max(g1 - jb, 0), hi = min(g1 + g3 - jb, 4);
  const bool any = (unsigned)(i - g0) < (unsigned)g2 && hi > lo;
  return any ? (1u << hi) - (1u << lo) : 0u;
}
// the goal-channel bits of two packed pixels (bytes 1 and 3 of the word) selected by bits 0 and 1 of m
__device__ __forceinline__ uint32_t goal_pair(uint32_t m, uint32_t gdiff) {
  const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)m, 0, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)m, 1, 1);
  return (gdiff & m0) | ((gdiff << 16) & m1);
}

#ifndef SRL_PLANE_CAP
#define SRL_PLANE_CAP 640    // plane / side slots staged in LDS per group of rocks (10 KB; a rock holds ~60): 8 - 10 rocks per group
#endif
#define SRL_PLANE_ROUNDS ((SRL_PLANE_CAP + SRL_RENDER_THREADS - 1) / SRL_RENDER_THREADS)

// LDS carve of srl_k_render (64 KB tile + 10 KB planes + 5 KB per-rock records at 128^2: two workgroups per CU)
struct RenderLds {
  float* tile;      // [res*res]
  float4* planes;   // [SRL_PLANE_CAP] per rock region: its up-facing planes, then its outline sides
  int* prange;      // [32][4] i0|i1<<16, j0|j1<<16, nir (item rows with spans; 0 = bounding-box items), items
  int* reg;         // [32][4] region base in its group, nup, nsil, slots before this rock (over all rocks)
  int* span;        // [32][16] per item row: first item << 8 | first column
  int* range;       // [32][16] per item row: plane_lo | plane_hi << 8 | side_lo << 16 | side_hi << 24
  uint32_t* rowmask;   // [8] bit i set: tile row i may hold a rock pixel (union of the rocks' row ranges)
  int* misc;        // [4] rocks in the first group
  float* pi;        // [512]  (pi and pu alias the plane staging area: used after the ray cast)
  float* pu;        // [512]
};

__host__ __device__ inline size_t render_lds_bytes(int res) {
  return sizeof(float) * (size_t)res * res + sizeof(float4) * SRL_PLANE_CAP +
         sizeof(int) * (4 + 4 + 2 * SRL_STAGE_SPANS) * SRL_MAX_BODIES + sizeof(uint32_t) * (8 + 4);
}

// one min / max sweep over planes [0, n) of a region for a lane's item of SRL_ITEM_ROWS x 2 pixels, this lane taking
// 4-plane batches s, s+S, ...
// z = fmaf(a, px, fmaf(b, py, c)) per pixel (the definition), evaluated two pixels at a
// time with packed fp32 FMAs (v_pk_fma_f32: IEEE fma per half, same bits as the scalar form).  The plane is
// fetched as one 16-byte LDS read (ds_read_b128: 4 LDS cycles per wave against 8 for a 12-byte read).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc[r] = (row r, column 0 | column 1); t = fma(b, y, c) is shared by the rows
template <bool UP>
__device__ __forceinline__ void plane_eval(const float4 q, const f32x2 py, const f32x2 (&vx)[SRL_ITEM_ROWS],
                                           f32x2 (&acc)[SRL_ITEM_ROWS]) {
  const f32x2 a = {q.x, q.x}, b = {q.y, q.y}, c = {q.z, q.z};
  const f32x2 t = __builtin_elementwise_fma(b, py, c);
#pragma unroll
  for (int r = 0; r < SRL_ITEM_ROWS; ++r) {
    const f32x2 z = __builtin_elementwise_fma(a, vx[r], t);
    if (UP) { acc[r].x = fminf(acc[r].x, z.x); acc[r].y = fminf(acc[r].y, z.y); }
    else { acc[r].x = fmaxf(acc[r].x, z.x); acc[r].y = fmaxf(acc[r].y, z.y); }
  }
}

template <bool UP>
__device__ __forceinline__ void plane_sweep(const float4* pl, int n, int s, int S, const f32x2 py,
                                            const f32x2 (&vx)[SRL_ITEM_ROWS], f32x2 (&acc)[SRL_ITEM_ROWS]) {
  if (n >= 4) {
    // ceil(n / 4) batches of four planes; batch k starts at plane min(4 k, n - 4), so the last one overlaps its
    // predecessor instead of leaving a remainder, and a trip takes two batches, the second clamped to the last batch
    // of the list (min / max are idempotent: a plane met twice changes nothing).  No selects, no remainder loop.
    const int nbt = (n + 3) >> 2, last = n - 4;
    for (int t = s; t < nbt; t += 2 * S) {
#ifdef SRL_ABL_NOPLANEREAD
      const int oa = 0, ob = 4;   // (diagnostic: the same eight planes every trip, fetched once)
#else
      const int oa = min(4 * t, last), ob = min(4 * (t + S), last);
#endif
      float4 q0 = pl[oa], q1 = pl[oa + 1], q2 = pl[oa + 2], q3 = pl[oa + 3];
      float4 r0 = pl[ob], r1 = pl[ob + 1], r2 = pl[ob + 2], r3 = pl[ob + 3];
#ifndef SRL_ABL_NOPLANEREAD
      asm volatile("" : "+v"(q0.w), "+v"(q1.w), "+v"(q2.w), "+v"(q3.w));   // keeps each fetch one 16-byte read
#endif
      plane_eval<UP>(q0, py, vx, acc); plane_eval<UP>(q1, py, vx, acc);
      plane_eval<UP>(q2, py, vx, acc); plane_eval<UP>(q3, py, vx, acc);
#ifndef SRL_ABL_NOPLANEREAD
      __builtin_amdgcn_sched_barrier(0);             // (the second batch arrives under the first one's arithmetic)
      asm volatile("" : "+v"(r0.w), "+v"(r1.w), "+v"(r2.w), "+v"(r3.w));
#endif
      plane_eval<UP>(r0, py, vx, acc); plane_eval<UP>(r1, py, vx, acc);
      plane_eval<UP>(r2, py, vx, acc); plane_eval<UP>(r3, py, vx, acc);
    }
  } else {
    for (int t = s; t < n; t += S) {
      float4 q = pl[t];
      asm volatile("" : "+v"(q.w));
      plane_eval<UP>(q, py, vx, acc);
    }
  }
}

// the kernel's output stores (H float4, packed observation uint2) are non-temporal: the 100 MB a launch writes pass through a
// 4 MB L2 per XCD once and are read by later kernels only in part (36.0 -> 34.95 us per launch; SRL_PLAIN_STORES: plain ones)
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int nt_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void out_f4(float* base, int g, float4 v) {
#ifndef SRL_PLAIN_STORES
  __builtin_nontemporal_store((nt_f4){v.x, v.y, v.z, v.w}, (nt_f4*)base + g);
#else
  ((float4*)base)[g] = v;
#endif
}
__device__ __forceinline__ void out_u2(uint8_t* base, int g, uint2 v) {
#ifndef SRL_PLAIN_STORES
  __builtin_nontemporal_store((nt_u2){v.x, v.y}, (nt_u2*)base + g);
#else
  ((uint2*)base)[g] = v;
#endif
}

// min with the lane a DPP control selects (0xB1: lane ^ 1, 0x4E: lane ^ 2 within quads; 0x141: mirror within half rows)
template <int CTRL>
__device__ __forceinline__ float dpp_min(float v) {
  return fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)));
}

// The in-wave stages of the halving tree (p[t] += p[t + s] for t < s, s = 32, 16, 8, 4, 2, 1; lane 0 ends with the sum): the
// partner's value comes by v_permlane32_swap / v_permlane16_swap (halves, rows of 16 lanes) and DPP row shifts instead of six
// ds_bpermute round trips (`__shfl_down`) — the same pairs added in the same order, so the same bits.
__device__ __forceinline__ float halving_sum(float v) {
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = v + __uint_as_float(r[1]);   // r[1] lanes 0..31 = lanes 32..63 of v
  }
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = v + __uint_as_float(r[1]);   // r[1] row 0 = row 1 of v (lanes 16..31)
  }
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x108, 0xf, 0xf, true));   // row_shl:8
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0xf, true));   // row_shl:4
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x102, 0xf, 0xf, true));   // row_shl:2
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));   // row_shl:1
  return v;
}

// inclusive prefix sum over lanes 0 .. 31 (integers): four DPP row shifts within the rows of 16 lanes (zero shifted in) and one
// row broadcast that adds lane 15 to row 1 — no ds_bpermute round trips in the prologue's chain
__device__ __forceinline__ int prefix32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
  return v;
}

// ------------------------------------------------------------------ staging as a kernel: one wave per (env, rock)
// The step path stages an env's rocks in the tail of its settle workgroup (settle.hip, stage.h) since round 5; this kernel
// serves the explicit-pose hook (srl_render_heightmap: poses_ext != nullptr) and handles whose records are out of date
// (after srl_set_body_state / srl_step_simulation, stackrl_hip.hip).  Same arithmetic: stage_compute.
extern "C" __global__ void __launch_bounds__(256) srl_k_stage(DevParams P, float4* __restrict__ stage, int slots,
    const float* __restrict__ poses_ext, const int32_t* __restrict__ mesh_ext, const int32_t* __restrict__ nb_ext) {
  __shared__ StageLds lds[4];
  const int e = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y * 4 + wave;
  if (b >= slots) return;                            // (wave-uniform; nothing below synchronises across waves)
  const bool ext = poses_ext != nullptr;
  // pose and mesh id of the slot are requested together with the rock count (slots past it hold stale or no data)
  q4 q; v3 xb; int m;
  if (ext) {
    const float* p = poses_ext + ((size_t)e * SRL_MAX_BODIES + b) * 7;
    xb = V(p[0], p[1], p[2]); q.x = p[3]; q.y = p[4]; q.z = p[5]; q.w = p[6];
    m = mesh_ext[(size_t)e * SRL_MAX_BODIES + b];
  } else {
    const float* gb = P.blob + (size_t)e * P.BLOB;
    xb = ld3(gb + P.OFF_X + 4 * b);
    const float* qq = gb + P.OFF_Q + 4 * b;
    q.x = qq[0]; q.y = qq[1]; q.z = qq[2]; q.w = qq[3];
    m = ((const int*)gb)[P.OFF_MESH + b];
  }
  const int nb = ext ? nb_ext[e] : P.hdr[e].nb;
  if (b >= nb) return;
  m = min(max(m, 0), P.n_mesh - 1);
  const MeshHdr mh = P.mh[m];
  StageArgs A;
  A.mp = P.mp; A.mt = P.mt; A.me = P.me; A.mv = P.mv; A.px = P.px; A.inv_px = P.inv_px; A.res = P.c.overhead_res;
  const StageLoads g = stage_request(A, mh, lane);
  stage_compute(A, lds[wave], stage + ((size_t)e * slots + b) * SRL_STAGE_STRIDE, xb, q, mh, g, lane);
}

// poses_ext != nullptr: test/profiling hook rendering explicit poses (srl_render_heightmap)
extern "C" __global__ void __launch_bounds__(SRL_RENDER_THREADS, 4)
srl_k_render(DevParams P, const float4* __restrict__ stage, int slots, uint8_t* __restrict__ obs_map,
             uint8_t* __restrict__ obs_obj, float* __restrict__ reward, uint8_t* __restrict__ done,
             const int32_t* __restrict__ nb_ext, float* __restrict__ height_ext) {
  extern __shared__ float4 lds_raw[];
#ifdef SRL_ABL_EMPTY
  if (P.px != 12345.0f) return;
#endif
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int res = P.c.overhead_res, npx = res * res;
  RenderLds L;
  L.tile = (float*)lds_raw;
  L.planes = (float4*)(L.tile + npx);
  L.prange = (int*)(L.planes + SRL_PLANE_CAP);
  L.reg = L.prange + 4 * SRL_MAX_BODIES;
  L.span = L.reg + 4 * SRL_MAX_BODIES;
  L.range = L.span + SRL_STAGE_SPANS * SRL_MAX_BODIES;
  L.rowmask = (uint32_t*)(L.range + SRL_STAGE_SPANS * SRL_MAX_BODIES);
  L.misc = (int*)(L.rowmask + 8);
  L.pi = (float*)L.planes;
  L.pu = L.pi + SRL_RENDER_THREADS;
  EnvHdr* h = &P.hdr[e];
  const float* gb = P.blob + (size_t)e * P.BLOB;
  const bool ext = nb_ext != nullptr;
#ifdef SRL_STAMPS
  long long _t0 = wall_clock64();
#endif
  // ---- prologue.  The record headers of every body slot are requested before the rock count is known (one memory round
  //      trip less); so are the words of the episode state the tail needs: their latency overlaps the ray cast.
  const float4* srec = stage + (size_t)e * slots * SRL_STAGE_STRIDE;
  int4 hd0 = make_int4(0, -65536, 0, 0), hd1 = make_int4(0, 0, 0, 0);
  if (tid < slots) {
    const float4 a = srec[(size_t)tid * SRL_STAGE_STRIDE], c = srec[(size_t)tid * SRL_STAGE_STRIDE + 1];
    hd0 = make_int4(__float_as_int(a.x), __float_as_int(a.y), __float_as_int(a.z), __float_as_int(a.w));
    hd1 = make_int4(__float_as_int(c.x), __float_as_int(c.y), __float_as_int(c.z), __float_as_int(c.w));
  }
  const int nb = ext ? nb_ext[e] : h->nb;
  int g0 = 0, g1 = 0, g2 = 0, g3 = 0, pending = -1, mode = 0, hdone = 0, left = 0;
  float prev_m[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (!ext) {
    g0 = h->goal[0]; g1 = h->goal[1]; g2 = h->goal[2]; g3 = h->goal[3]; pending = h->pending;
    mode = h->mode; hdone = h->done; left = h->list_pos;
#pragma unroll
    for (int k = 0; k < 4; ++k) prev_m[k] = h->prev_metric[k];
  }
  const int rr = P.c.object_res * P.c.object_res * P.n_orient;   // all observable orientations of the pending rock
  // (the tile is zeroed where a rock may write — the groups of the rows some bounding box reaches — in the early-store loop
  //  below, whose thread-to-group map the epilogue shares; a block barrier lies between that loop and the ray cast)
  if (tid < 8) L.rowmask[tid] = 0u;                  // (wave 0, like the atomicOr below: LDS keeps a wave's order)
  if (tid < SRL_MAX_BODIES) {
    // a rock's region holds its up-facing planes and the sides of its outline (a rock outside the window has neither)
    const int nt = tid < nb ? hd0.z + hd0.w : 0;
    int pre = nt;   // inclusive prefix of the region sizes over the rocks (lanes 0..31 of wave 0)
    pre = prefix32(pre);
    const int be0 = __popcll(__ballot(tid < nb && pre <= SRL_PLANE_CAP));   // rocks whose planes fit the first group
    if (tid == 0) L.misc[0] = be0;
    if (tid < nb) {
      L.prange[4 * tid + 0] = hd0.x; L.prange[4 * tid + 1] = hd0.y; L.prange[4 * tid + 2] = hd1.y; L.prange[4 * tid + 3] = hd1.x;
      // cursors as for the first group (later groups reset theirs)
      L.reg[4 * tid + 0] = pre - nt; L.reg[4 * tid + 1] = hd0.z; L.reg[4 * tid + 2] = hd0.w; L.reg[4 * tid + 3] = pre - nt;
      if (hd1.x > 0) {
        const int i0 = hd0.x & 0xffff, i1 = hd0.x >> 16;
        for (int w = i0 >> 5; w <= i1 >> 5; ++w) {
          const int lo = max(i0 - 32 * w, 0), hi = min(i1 - 32 * w, 31);
          atomicOr(&L.rowmask[w], (0xffffffffu >> (31 - hi)) & (0xffffffffu << lo));
        }
      }
    }
  }
  // the span and range tables of the rocks: eight 16-byte words each, one per thread
  if (tid < 8 * nb) {
    const int b = tid >> 3, k = tid & 7;
    const float4 v = srec[(size_t)b * SRL_STAGE_STRIDE + 2 + k];
    if (k < 4) ((float4*)L.span)[4 * b + k] = v; else ((float4*)L.range)[4 * b + k - 4] = v;
  }
  // epilogue constants
  const float nearp = SRL_FAR - P.c.max_z;
  const float gz = P.goal_z;
  const uint32_t gbyte = P.gbyte, zbyte = P.zbyte, b_empty = P.b_empty;   // (DevParams: evaluated once on the device)
  const float h_empty = P.h_empty;
  float* Hout = ext ? height_ext + (size_t)e * npx : P.H + (size_t)e * npx;
  uint8_t* om = ext ? nullptr : obs_map + (size_t)e * npx * 2;
  const int ngroups4 = npx / 4, nrounds = (ngroups4 + SRL_RENDER_THREADS - 1) / SRL_RENDER_THREADS;
  // pixel-group walk of a thread: group g = tid + 512 k holds pixels 4 g .. 4 g + 3 = row i, columns jb .. jb + 3
  // (res is a multiple of 8); from one round to the next the group advances by di rows and dj columns
  const int walk_di = P.walk_di, walk_dj = P.walk_dj;
  const int walk_i0 = (int)__umulhi((uint32_t)(4 * tid), P.res_magic), walk_j0 = 4 * tid - walk_i0 * res;
  // res divides 4 * 512 (64, 128, 256 ...): the walk never changes columns, so the column part of the goal test is
  // a per-thread constant: bit t of colmask = column walk_j0 + t lies in the goal rectangle's column range
  // (maps of fewer than 4 x 512 pixels take the general walk: its rounds test the group index against the map size)
  const bool aligned = walk_dj == 0 && ngroups4 >= SRL_RENDER_THREADS;
  uint32_t colmask = 0u;
  const uint32_t gdiff = (gbyte ^ zbyte) << 8, zpair = (zbyte << 8) | (zbyte << 24);
  {
    const int lo = max(g1 - walk_j0, 0), hi = min(g1 + g3 - walk_j0, 4);
    colmask = hi > lo ? (1u << hi) - (1u << lo) : 0u;
  }
  // ---- object observation (O2 from the per-mesh cache; empty map when nothing is pending): its inputs are known here,
  //      so it leaves now and drains under everything else
#ifndef SRL_ABL_NOOBJ
  if (!ext) {
    // the pending rock's maps, or with ordering freedom (observer.py:310-327) those of the rocks still unplaced, in list
    // order, then empty maps; bytes from the per-mesh cache, four pixels per lane (the map size is a multiple of 4)
    const int shown = P.c.ordering_freedom ? P.c.episode_length : 1;
    uint32_t* oo = (uint32_t*)(obs_obj + (size_t)e * rr * shown);
    const uint32_t eb4 = P.obj_empty_byte * 0x01010101u;
    for (int k = 0; k < shown; ++k) {
      const int m = P.c.ordering_freedom ? (k < left ? h->ids[k] : -1) : pending;
      const uint32_t* src = (const uint32_t*)(P.objmap_u8 + (size_t)(m < 0 ? 0 : m) * rr);
      for (int idx = tid; idx < rr / 4; idx += SRL_RENDER_THREADS) oo[(size_t)k * (rr / 4) + idx] = m >= 0 ? src[idx] : eb4;
    }
  }
#endif
  __syncthreads();
  RSTAMP(0);
  // ---- groups of rocks whose planes fit the staging area
  int bs = 0, be = L.misc[0];   // (be >= 1 when nb >= 1: a mesh has at most SRL_MAX_TRIS + 2 <= SRL_PLANE_CAP slots)
  uint32_t cov = 0u, goalm = 0u;
  bool first = true;
  do {
    // (a) the group's planes and outline sides from their records (srl_k_stage) into the regions, the rocks handed to the
    //     8 waves round robin, lanes over a rock's slots: the first 64 slots of a wave's first rock are requested now and
    //     written to LDS after the early stores below (their latency hides under those), the rest follows there.
    //     (Requesting them two barriers earlier, from the prologue, was measured: no gain.)
    const int pb = bs + wave;
    int pbase = 0, pcnt = 0;
    if (pb < be) { pbase = L.reg[4 * pb + 0]; pcnt = L.reg[4 * pb + 1] + L.reg[4 * pb + 2]; }
    float4 pv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#ifdef SRL_ABL_NOSTAGE
    if (P.px == 12345.0f)
#endif
    if (lane < pcnt) pv = srec[(size_t)pb * SRL_STAGE_STRIDE + SRL_STAGE_HDR + lane];
    RSTAMP(1);
    // (d) first trip: rows no rock reaches hold the empty-pixel constants; their H / observation bytes leave
    //     now, so that HBM drains them while the ray cast computes.  cov bit k: this thread's k-th pixel group
    //     lies in a row some rock may reach (written by the epilogue).
    if (first) {
      const float4 he4 = make_float4(h_empty, h_empty, h_empty, h_empty);
      const uint32_t epair = b_empty | (b_empty << 16);
      if (aligned) {
        // every round keeps the thread's four columns and moves walk_di rows down: the column part of the goal test
        // and the two possible observation words are fixed, a round costs a row test and a row-mask bit (the mask's four
        // words — res <= 128 on this path — are read once)
        const uint4 rm4 = *(const uint4*)L.rowmask;
        const unsigned long long rlo = rm4.x | ((unsigned long long)rm4.y << 32), rhi = rm4.z | ((unsigned long long)rm4.w << 32);
        const uint32_t w_out = epair | zpair, w_in_lo = w_out ^ goal_pair(colmask, gdiff), w_in_hi = w_out ^ goal_pair(colmask >> 2, gdiff);
        for (int k = 0; k < nrounds; ++k) {
          const int g = tid + k * SRL_RENDER_THREADS, i = walk_i0 + k * walk_di;
          const bool rowin = (unsigned)(i - g0) < (unsigned)g2;
#ifndef SRL_ABL_NOGOAL
          if (rowin && colmask) goalm |= 1u << k;
#endif
          if (((i < 64 ? rlo : rhi) >> (i & 63)) & 1ull) { cov |= 1u << k; ((float4*)L.tile)[g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
          else {
#ifndef SRL_ABL_NOSTORE
            out_f4(Hout, g, he4);
            if (om) out_u2(om, g, make_uint2(rowin ? w_in_lo : w_out, rowin ? w_in_hi : w_out));
#endif
          }
        }
      } else {
      int i = walk_i0, jb = walk_j0;
      for (int k = 0; k < nrounds; ++k) {
        const int g = tid + k * SRL_RENDER_THREADS;
        if (g >= ngroups4) break;
        // groups that touch the goal rectangle: their empty pixels still add the goal height to the union sum
#ifndef SRL_ABL_NOGOAL
        const uint32_t inm = goal_mask4(i, jb, g0, g2, g1, g3);
#else
        const uint32_t inm = 0u;
#endif
        if (inm) goalm |= 1u << k;
        if ((L.rowmask[i >> 5] >> (i & 31)) & 1u) { cov |= 1u << k; ((float4*)L.tile)[g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
        else {
#ifndef SRL_ABL_NOSTORE
          out_f4(Hout, g, he4);
          if (om) out_u2(om, g, make_uint2((epair | zpair) ^ goal_pair(inm, gdiff), (epair | zpair) ^ goal_pair(inm >> 2, gdiff)));
#endif
        }
        jb += walk_dj; i += walk_di;
        if (jb >= res) { jb -= res; ++i; }
      }
      }
      RSTAMP(2);
    }
    // the group's slots into their regions
#ifdef SRL_ABL_NOSTAGE
    if (P.px == 12345.0f)
#endif
    {
      if (lane < pcnt) L.planes[pbase + lane] = pv;
      for (int b = pb; b < be; b += SRL_RENDER_THREADS / 64) {
        const int base = L.reg[4 * b + 0], cnt = L.reg[4 * b + 1] + L.reg[4 * b + 2];
        const float4* src = srec + (size_t)b * SRL_STAGE_STRIDE + SRL_STAGE_HDR;
        for (int k = (b == pb ? 64 : 0) + lane; k < cnt; k += 64) L.planes[base + k] = src[k];
      }
    }
    __syncthreads();
    // (e) ray cast: lanes over the flattened (rock, item of SRL_ITEM_ROWS x 2 pixels) list; when the list is short each item
    //     is shared by S adjacent lanes that split the planes and combine with shuffles
#ifdef SRL_ABL_NOCAST
    if (be > bs && P.px == 12345.0f) {
#else
    if (be > bs) {
#endif
      // item counts of the group's rocks as running sums (registers, one LDS round trip); groups of more
      // than 8 rocks fall back to walking the list
      const bool few = be - bs <= 8;
      int pre[8];
      int total = 0;
      if (few) {
#pragma unroll
        for (int r = 0; r < 8; ++r) pre[r] = bs + r < be ? L.prange[4 * (bs + r) + 3] : 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) { total += pre[r]; pre[r] = total; }
      } else {
        for (int b = bs; b < be; ++b) total += L.prange[4 * b + 3];
      }
      int lg = 0;
      while (lg < 3 && (total << (lg + 1)) <= SRL_RENDER_THREADS) ++lg;
      const int S = 1 << lg, s = tid & (S - 1);
      int wb = bs, wis = 0, wnib = few ? 0 : L.prange[4 * bs + 3];
      for (int it = tid >> lg; it < total; it += SRL_RENDER_THREADS >> lg) {
        int b, is;
        if (few) {
          b = bs; is = 0;
#pragma unroll
          for (int r = 0; r < 7; ++r) if (it >= pre[r]) { b = bs + r + 1; is = pre[r]; }
        } else {
          while (it >= wis + wnib) { wis += wnib; ++wb; wnib = L.prange[4 * wb + 3]; }
          b = wb; is = wis;
        }
        const int4 pr = ((const int4*)L.prange)[b], rg = ((const int4*)L.reg)[b];
        const int nir = pr.z, jj = pr.y, ii = pr.x;
        const int j0 = jj & 0xffff, j1 = jj >> 16, i1 = ii >> 16;
        const int p = it - is;
        int di, j;
        if (nir > 0) {
          // item row = the last one whose first item is <= p (rows the outline does not reach share their successor's first
          // item, rows past the last start at the rock's item count); the row's first column is in the low byte
          const int4* sp = (const int4*)(L.span + SRL_STAGE_SPANS * b);
          const int4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
          const int sv[SRL_STAGE_SPANS] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w, s3.x, s3.y, s3.z, s3.w};
          const int key = (p << 8) | 0xff;
          int w = sv[0];
          di = 0;
#pragma unroll
          for (int r = 1; r < SRL_STAGE_SPANS; ++r) if (key >= sv[r]) { w = sv[r]; di = r; }
          j = (w & 0xff) + 2 * (p - (w >> 8));
        } else {
          // items fill the bounding box.  p / w2 without the integer-division sequence: (p + 0.5) / w2 stays at least
          // 0.5 / w2 >= 1 / 256 away from an integer, far more than the error of the reciprocal (p < 2^14, w2 <= 128)
          const int w2 = (j1 - j0 + 2) >> 1;
          di = (int)(((float)p + 0.5f) * __builtin_amdgcn_rcpf((float)w2));
          j = j0 + 2 * (p - di * w2);
        }
        const int i = (ii & 0xffff) + SRL_ITEM_ROWS * di;
        const bool col2 = j + 1 <= j1;
        const f32x2 py = {((float)j + 0.5f) * P.px, ((float)(j + 1) + 0.5f) * P.px};
        f32x2 vx[SRL_ITEM_ROWS], hh[SRL_ITEM_ROWS], ll[SRL_ITEM_ROWS];
#pragma unroll
        for (int r = 0; r < SRL_ITEM_ROWS; ++r) {
          const float x = ((float)(i + r) + 0.5f) * P.px;
          vx[r].x = x; vx[r].y = x; hh[r].x = 1e30f; hh[r].y = 1e30f; ll[r].x = 1e30f; ll[r].y = 1e30f;
        }
        // z_hi = min over the up-facing planes; inside the outline iff the smallest side function is >= 0
        // the item row's ranges of the two lists (srl_k_stage): the faces / outline edges whose x extent reaches the row
        const uint32_t rw = nir > 0 ? (uint32_t)L.range[SRL_STAGE_SPANS * b + di] : ((uint32_t)rg.y << 8) | ((uint32_t)rg.z << 24);
        const int plo = rw & 0xff, pn = ((rw >> 8) & 0xff) - plo, slo = (rw >> 16) & 0xff, sn = (int)(rw >> 24) - slo;
        plane_sweep<true>(L.planes + rg.x + plo, pn, s, S, py, vx, hh);
        plane_sweep<true>(L.planes + rg.x + rg.y + slo, sn, s, S, py, vx, ll);
        const bool lists = pn > 0 && sn > 0;           // (a row the lists do not reach holds no rock pixel)
        // the S lanes of an item (adjacent, aligned) combine their partial minima with DPP moves folded into the min:
        // within quads (lanes ^ 1, ^ 2), then across the two quads of a half row (mirror) — no LDS traffic
        if (S >= 2) {
#pragma unroll
          for (int r = 0; r < SRL_ITEM_ROWS; ++r) {
            hh[r].x = dpp_min<0xB1>(hh[r].x); hh[r].y = dpp_min<0xB1>(hh[r].y);
            ll[r].x = dpp_min<0xB1>(ll[r].x); ll[r].y = dpp_min<0xB1>(ll[r].y);
          }
        }
        if (S >= 4) {
#pragma unroll
          for (int r = 0; r < SRL_ITEM_ROWS; ++r) {
            hh[r].x = dpp_min<0x4E>(hh[r].x); hh[r].y = dpp_min<0x4E>(hh[r].y);
            ll[r].x = dpp_min<0x4E>(ll[r].x); ll[r].y = dpp_min<0x4E>(ll[r].y);
          }
        }
        if (S >= 8) {
#pragma unroll
          for (int r = 0; r < SRL_ITEM_ROWS; ++r) {
            hh[r].x = dpp_min<0x141>(hh[r].x); hh[r].y = dpp_min<0x141>(hh[r].y);
            ll[r].x = dpp_min<0x141>(ll[r].x); ll[r].y = dpp_min<0x141>(ll[r].y);
          }
        }
        if (s == 0) {
          // the tile holds codec-table rows, not heights: row(z) = codec_n - index of t = fl(FAR - z) is monotone in z,
          // so the max over rocks commutes with the look-up and the epilogue needs no arithmetic per pixel (0 = no rock)
          int* t0p = (int*)&L.tile[i * res + j];
#pragma unroll
          for (int r = 0; r < SRL_ITEM_ROWS; ++r) {
            const bool rowok = i + r <= i1;
            if (lists && rowok && ll[r].x >= 0.0f && hh[r].x > 0.0f) atomicMax(t0p + r * res, codec_row(hh[r].x, nearp, P.codec_n));
            if (lists && rowok && col2 && ll[r].y >= 0.0f && hh[r].y > 0.0f) atomicMax(t0p + r * res + 1, codec_row(hh[r].y, nearp, P.codec_n));
          }
        }
      }
    }
    __syncthreads();
    RSTAMP(3);
    first = false;
    bs = be;
    if (bs < nb) {   // next group: its extent, then its region cursors
      const int p0 = L.reg[4 * bs + 3];
      while (be < nb && L.reg[4 * be + 3] + L.reg[4 * be + 1] + L.reg[4 * be + 2] - p0 <= SRL_PLANE_CAP) ++be;
      __syncthreads();                               // (every thread has read the old cursors)
      if (tid >= bs && tid < be) L.reg[4 * tid + 0] = L.reg[4 * tid + 3] - p0;
      __syncthreads();
    }
  } while (bs < nb);
  RSTAMP(4);
  // ---- epilogue: depth codec, H out, uint8 pack, IoU partial sums (4 pixels per thread per round, fixed order).
  //      The kernel is bound by VALU issue, so the pass is branch-free per pixel: the codec is one 8-byte look-up
  //      (DevParams::codec; a pixel without a rock reads the table's last entry, the empty-pixel constants), the goal
  //      terms are selected with bit masks.
  float spi = 0.0f, spu = 0.0f;
  {
    // An empty pixel outside the goal adds max(h_empty, 0) to the union sum and nothing else; when that is +0 (it is
    // whenever FAR (FAR - max_z) is a float32, e.g. the default geometry) the addition is the identity, bit for bit
    // (the partial sums never hold -0), so rounds whose group lies in a row no rock reaches and outside the goal
    // rectangle are skipped altogether.  todo bit k: round k must be visited.
    const uint32_t todo = fmaxf(h_empty, 0.0f) == 0.0f ? (cov | goalm) : 0xffffffffu;
    const uint2* __restrict__ ctab = P.codec;
    const uint32_t gp_lo = goal_pair(colmask, gdiff), gp_hi = goal_pair(colmask >> 2, gdiff);
    // what the four pixels of a group no rock reaches add to the two sums in a goal row (the columns are fixed per
    // thread when `aligned`): the very terms the general path below computes from the table's empty-pixel entry
    float cu_in[4], ci_in[4];
    const float cu_out = fmaxf(h_empty, 0.0f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int m = __builtin_amdgcn_sbfe((int)colmask, t, 1);
      cu_in[t] = fmaxf(h_empty, __int_as_float(__float_as_int(gz) & m));
      ci_in[t] = __int_as_float(__float_as_int(fminf(h_empty, gz)) & m);
    }
    int i = walk_i0, jb = walk_j0;
    for (int k = 0; k < nrounds; ++k) {
      const int g = tid + k * SRL_RENDER_THREADS;
      if (((todo >> k) & 1u) && g < ngroups4) {
        const bool covg = (cov >> k) & 1u;
#ifndef SRL_ABL_NOCONSTROUND
        if (aligned && __builtin_amdgcn_ballot_w64(covg) == 0ull) {
          // no group of the wave holds a rock pixel this round (their H / observation bytes left with the early stores):
          // the sums take per-thread constants — no table look-ups, no stores
          const bool rowin = (unsigned)(i - g0) < (unsigned)g2;
#pragma unroll
          for (int t = 0; t < 4; ++t) { spu += rowin ? cu_in[t] : cu_out; spi += rowin ? ci_in[t] : 0.0f; }
          jb += walk_dj; i += walk_di;
          if (jb >= res) { jb -= res; ++i; }
          continue;
        }
#endif
        uint4 r4 = make_uint4(0u, 0u, 0u, 0u);   // codec-table rows of the four pixels (0: no rock)
        if (covg) r4 = ((const uint4*)L.tile)[g];
        const uint32_t rw[4] = {r4.x, r4.y, r4.z, r4.w};
        float hv[4]; uint32_t hb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint2 en = *(const uint2*)((const char*)ctab + (rw[t] << 3));   // 32-bit offset from a uniform base
          hv[t] = __uint_as_float(en.x); hb[t] = en.y;
        }
        // rewarder.py:297-307: goal pixels add min / max(h, goal) to the two sums, the others max(h, 0) to the union
        // only (x + 0 = x bit for bit here: the partial sums never hold -0); m = all ones on a goal pixel
        uint32_t glo = 0u, ghi = 0u;   // goal-channel bits of the two observation words
        const bool rowin = (unsigned)(i - g0) < (unsigned)g2;
        if (aligned && __builtin_amdgcn_ballot_w64(rowin && colmask != 0u) == 0ull) {
          // none of the wave's groups of this round touches the goal (a wave covers whole rows): union sum only
#pragma unroll
          for (int t = 0; t < 4; ++t) spu += fmaxf(hv[t], 0.0f);
        } else {
          const uint32_t inm = aligned ? (rowin ? colmask : 0u) : goal_mask4(i, jb, g0, g2, g1, g3);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int m = __builtin_amdgcn_sbfe((int)inm, t, 1);
            spu += fmaxf(hv[t], __int_as_float(__float_as_int(gz) & m));
            spi += __int_as_float(__float_as_int(fminf(hv[t], gz)) & m);
          }
          if (aligned) { glo = rowin ? gp_lo : 0u; ghi = rowin ? gp_hi : 0u; }
          else { glo = goal_pair(inm, gdiff); ghi = goal_pair(inm >> 2, gdiff); }
        }
#ifndef SRL_ABL_NOSTORE
        if (covg) {
          out_f4(Hout, g, make_float4(hv[0], hv[1], hv[2], hv[3]));
          if (om) out_u2(om, g, make_uint2((hb[0] | (hb[1] << 16) | zpair) ^ glo, (hb[2] | (hb[3] << 16) | zpair) ^ ghi));
        }
#else
        if (covg && hv[0] == 12345.0f && hb[0] == 77u) ((float4*)Hout)[g] = make_float4(hv[0], hv[1], hv[2], hv[3]);
#endif
      }
      jb += walk_dj; i += walk_di;
      if (jb >= res) { jb -= res; ++i; }
    }
  }
  RSTAMP(5);
  if (ext) return;
  // ---- halving tree over the 512 partials: cross-wave stages through LDS, in-wave stages by shuffles
#ifdef SRL_ABL_NOTAIL
  if (P.px != 12345.0f) return;
#endif
  //      (p[t] += p[t + 256], t < 256; p[t] += p[t + 128], t < 128; p[t] += p[t + 64], t < 64; then shuffles): the three
  //      cross-wave stages only ever combine lane t of the eight waves, so wave 0 reads the eight partials after ONE barrier
  //      and adds them in the tree's order — the same sums, bit for bit, with two barriers fewer
  L.pi[tid] = spi; L.pu[tid] = spu;
  __syncthreads();
  float sum_i = 0.0f, sum_u = 0.0f;
  if (tid < 64) {
    float qi[8], qu[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { qi[k] = L.pi[tid + 64 * k]; qu[k] = L.pu[tid + 64 * k]; }
    sum_i = ((qi[0] + qi[4]) + (qi[2] + qi[6])) + ((qi[1] + qi[5]) + (qi[3] + qi[7]));
    sum_u = ((qu[0] + qu[4]) + (qu[2] + qu[6])) + ((qu[1] + qu[5]) + (qu[3] + qu[7]));
    sum_i = halving_sum(sum_i); sum_u = halving_sum(sum_u);   // p[t] += p[t + s], t < s, s = 32 ... 1: lane 0 holds the sum
  }
  // ---- K5: reward = scale * (metric_t - metric_{t-1})  (rewarder.py:176-179)
  //      'all' (rewarder.py:157-158): the four metrics at once, reward[e][4]; 'eval' (rewarder.py:147-156): reward[e][2] = the
  //      IoU reward and the change of the average discount ('AD', memory slot -1 = 3, not scaled)
  if (tid == 0) {
    const int K = P.c.metric == SRL_METRIC_ALL ? 4 : P.c.metric == SRL_METRIC_EVAL ? 2 : 1;
    float* rw = reward + (size_t)e * K;
    if (mode == 0) {
      const float iou = sum_i / sum_u, orr = sum_i / ((float)(g2 * g3) * gz);
      if (P.c.metric == SRL_METRIC_ALL) {
        const float mv[4] = {iou, orr, discounted_metric(P, h, gb, SRL_METRIC_DIOU), discounted_metric(P, h, gb, SRL_METRIC_DOR)};
#pragma unroll
        for (int m = 0; m < 4; ++m) { rw[m] = (mv[m] - prev_m[m]) * P.scale; h->prev_metric[m] = mv[m]; }
      } else if (P.c.metric == SRL_METRIC_EVAL) {
        rw[0] = (iou - prev_m[0]) * P.scale; h->prev_metric[0] = iou;
        const float ad = average_discount(P, h, gb);
        rw[1] = ad - prev_m[3]; h->prev_metric[3] = ad;
      } else {
        float mv;
        if (P.c.metric == SRL_METRIC_IOU) mv = iou;
        else if (P.c.metric == SRL_METRIC_OR) mv = orr;
        else mv = discounted_metric(P, h, gb, P.c.metric);
        float pm = prev_m[0];                        // (requested in the prologue; a constant index keeps it in registers)
#pragma unroll
        for (int k = 1; k < 4; ++k) if (P.c.metric == k) pm = prev_m[k];
        rw[0] = (mv - pm) * P.scale;
        h->prev_metric[P.c.metric] = mv;
      }
      done[e] = (uint8_t)hdone;
    } else {   // reset step (env.py:235-236) or rejected action
      for (int m = 0; m < K; ++m) rw[m] = 0.0f;
      done[e] = 0;
    }
  }
  RSTAMP(6);
}

// K3: underside map of one mesh at the spawn pose, in one observable orientation (blockIdx.y; Stack-v0 has only the
// identity).  One workgroup per (mesh, orientation).  The rock turns about its link-frame origin, which sits at the
// centre of the map (observer.py:143-164: the camera looks at the object pose).
extern "C" __global__ void __launch_bounds__(256) srl_k_objmap(DevParams P, float* __restrict__ out, uint8_t* __restrict__ out_u8) {
  __shared__ float4 planes[SRL_MAX_TRIS];
  const int m = blockIdx.x, oi = blockIdx.y, tid = threadIdx.x;
  const int r = P.c.object_res;
  const float half = P.c.object_max_dimension * 0.5f;
  const MeshHdr mh = P.mh[m];
  m3 R;
  v3 xs;   // COM in map coordinates: link frame shifted to start at 0
  if (P.n_orient == 1) {
#pragma unroll
    for (int k = 0; k < 9; ++k) R.m[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    xs = V(mh.cx + half, mh.cy + half, mh.cz);
  } else {
    q4 q; q.x = P.orient_q[oi][0]; q.y = P.orient_q[oi][1]; q.z = P.orient_q[oi][2]; q.w = P.orient_q[oi][3];
    R = quat_to_mat(q);
    xs = mmul(R, V(mh.cx, mh.cy, mh.cz)) + V(half, half, 0.0f);
  }
  for (int t = tid; t < mh.nt; t += 256) planes[t] = make_rplane(P.mp[mh.to + t], R, xs);
  __syncthreads();
  const float nearp = SRL_FAR - half, farp = SRL_FAR + half;
  for (int k = tid; k < r * r; k += 256) {
    int i = k / r, j = k - i * r;
    float px = ((float)i + 0.5f) * P.px, py = ((float)j + 0.5f) * P.px;
    float lo, hi;
    float z = 1e30f;
    if (ray_cast(planes, mh.nt, px, py, lo, hi)) z = lo;
    float d = z > 1e29f ? 1.0f : depth_encode(SRL_FAR + z, nearp, farp);
    const float ev = elev_object(P, d);
    out[((size_t)m * P.n_orient + oi) * r * r + k] = ev;
    out_u8[((size_t)m * P.n_orient + oi) * r * r + k] = (uint8_t)((ev * 255.0f) / fmaxf(P.c.max_z, P.c.object_max_dimension));   // env.py:171-172
  }
}
