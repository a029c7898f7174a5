// render.hip — K2 (overhead height map) + K5 (reward) + observation packing, and K3 (object map).
//
// Renderer definition: convex ray cast.  A rock is the intersection of its face
// half-spaces; along the vertical line through a pixel centre the hull spans [z_lo, z_hi] with
// z_hi = min over up-facing planes and z_lo = max over down-facing planes.  The overhead camera
// (observer.py:252-260) sees z_hi, the object camera from below (observer.py:262-277) sees z_lo.
//
// K2: one 512-thread workgroup per env.  The res x res tile lives in LDS (64 KB at 128^2 -> 2
// workgroups = 16 waves per CU).  Rocks are visited one after the other; for each rock all lanes take
// pixels of its pixel bounding box (lane = pixel, fully regular, no atomics) and loop over the rock's
// planes, which are staged in LDS (double buffered, one barrier per rock) and read as wave-uniform
// broadcasts.  One epilogue pass applies the reference's depth codec (observer.py:259-260), streams out
// H (16 B per lane, 1 KB contiguous per wave store), the packed uint8 observation (env.py:171-172,
// :228-231) and accumulates the IoU sums (rewarder.py:297-307) in the fixed order DESIGN.md defines.
// Empty pixels (the majority) skip the codec arithmetic.  HBM traffic per env step is the algorithmic
// 6*res^2 + 5*r^2 bytes out plus ~1.4 KB per rock of mesh/pose data in (L2-resident pool).
//
// K3: per-mesh underside map rendered once at srl_load_meshes and cached: it depends only on the mesh
// (spawn orientation is the identity, env.py:120-121).
#include "srl_device.h"
#include "srl_kernels.h"

#define SRL_RENDER_THREADS 512

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

// world-frame render plane of one face: z = a x + b y + c; w = 0 up-facing (z_hi = min), 1 down-facing
// (z_lo = max).  |n_z| is clamped to >= 1e-6: a vertical face becomes a plane of enormous slope that never
// limits z on its inner side and empties the interval on its outer side.
__device__ __forceinline__ float4 make_rplane(float4 pl, const m3& R, v3 x) {
  v3 nw = mmul(R, V(pl.x, pl.y, pl.z));
  float dw = pl.w + dot(nw, x);
  float nz = nw.z;
  int type;
  if (nz >= 0.0f) { if (nz < 1e-6f) nz = 1e-6f; type = 0; }
  else { if (nz > -1e-6f) nz = -1e-6f; type = 1; }
  return make_float4(-nw.x / nz, -nw.y / nz, dw / nz, __int_as_float(type));
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

__device__ __forceinline__ bool pixel_range(float lo, float hi, float inv_px, int res, int& i0, int& i1) {
  float f0 = ceilf(lo * inv_px - 0.5f), f1 = floorf(hi * inv_px - 0.5f);
  if (f0 < 0.0f) f0 = 0.0f;
  if (f1 > (float)(res - 1)) f1 = (float)(res - 1);
  if (f1 < f0) return false;
  i0 = (int)f0; i1 = (int)f1;
  return true;
}

// Rewarder.call (rewarder.py:162-179) for the discounted metrics, thread 0 only
__device__ float discounted_metric(const DevParams& P, const EnvHdr* h, const float* gb) {
  float pmax = (float)P.c.object_res * P.px;   // rewarder.py:126
  float omax = 3.14159265358979f;
  float r = 0.0f; int nout = 0;
  for (int b = 0; b < h->nb; ++b) {
    v3 x = ld3(gb + P.OFF_X + 4 * b);
    float fu = floorf(x.x / P.px), fv = floorf(x.y / P.px);   // xy_to_pixel (observer.py:388-390)
    bool in = fu >= (float)h->goal[0] && fv >= (float)h->goal[1] && fu < (float)(h->goal[0] + h->goal[2]) &&
              fv < (float)(h->goal[1] + h->goal[3]);
    if (!in) { nout++; continue; }
    v3 dp = ld3(gb + P.OFF_PX + 4 * b) - x;
    float perr = sqrtf(dot(dp, dp));
    const float* a = gb + P.OFF_PQ + 4 * b;
    const float* q = gb + P.OFF_Q + 4 * b;
    float dw = fabsf((a[0] * q[0] + a[1] * q[1]) + (a[2] * q[2] + a[3] * q[3]));
    float oerr = 2.0f * srl_acosf(fminf(dw, 1.0f));
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
    r = r + disc;
  }
  if (P.c.metric == SRL_METRIC_DOR) return r / (float)P.c.episode_length;
  return r / (float)(P.c.episode_length + nout);
}

#define SRL_PLANE_CAP 512    // planes staged in LDS per group of rocks (8 KB); = SRL_RENDER_THREADS

// LDS carve of srl_k_render
struct RenderLds {
  float* tile;      // [res*res]
  float4* planes;   // [SRL_PLANE_CAP] per rock region: up-facing from the front, down-facing from the back
  float* sx;        // [32][3]
  float* sR;        // [32][9]
  int* mhdr;        // [32][4] vo, nv, to, nt
  uint32_t* bbox;   // [32][4] ordered-uint xmin, xmax, ymin, ymax
  int* prange;      // [32][4] i0|i1<<16, j0|j1<<16, w2 (quads per row), quads
  int* reg;         // [32][4] region base, up cursor, down cursor (exclusive, counts down), -
  uint32_t* rowmask;   // [8] bit i set: tile row i may hold a rock pixel (union of the rocks' row ranges)
  float* pi;        // [512]  (aliases the plane staging area: used after the ray cast)
  float* pu;        // [512]
};

__host__ __device__ inline size_t render_lds_bytes(int res) {
  return sizeof(float) * (size_t)res * res + sizeof(float4) * SRL_PLANE_CAP +
         sizeof(float) * (3 + 9 + 4 + 4 + 4 + 4) * SRL_MAX_BODIES + sizeof(uint32_t) * 8;
}

// one min / max sweep over planes [0, n) of a region, this lane taking 4-plane batches s, s+S, ...
template <bool UP>
__device__ __forceinline__ void plane_sweep(const float4* pl, int n, int s, int S, float px0, float px1, float py0,
                                            float py1, float& z00, float& z01, float& z10, float& z11) {
  const int nb4 = n >> 2;
  for (int kb = s; kb < nb4; kb += S) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 q = pl[4 * kb + u];
      const float t0 = fmaf(q.y, py0, q.z), t1 = fmaf(q.y, py1, q.z);
      if (UP) {
        z00 = fminf(z00, fmaf(q.x, px0, t0)); z01 = fminf(z01, fmaf(q.x, px0, t1));
        z10 = fminf(z10, fmaf(q.x, px1, t0)); z11 = fminf(z11, fmaf(q.x, px1, t1));
      } else {
        z00 = fmaxf(z00, fmaf(q.x, px0, t0)); z01 = fmaxf(z01, fmaf(q.x, px0, t1));
        z10 = fmaxf(z10, fmaf(q.x, px1, t0)); z11 = fmaxf(z11, fmaf(q.x, px1, t1));
      }
    }
  }
  for (int t = 4 * nb4 + s; t < n; t += S) {
    const float4 q = pl[t];
    const float t0 = fmaf(q.y, py0, q.z), t1 = fmaf(q.y, py1, q.z);
    if (UP) {
      z00 = fminf(z00, fmaf(q.x, px0, t0)); z01 = fminf(z01, fmaf(q.x, px0, t1));
      z10 = fminf(z10, fmaf(q.x, px1, t0)); z11 = fminf(z11, fmaf(q.x, px1, t1));
    } else {
      z00 = fmaxf(z00, fmaf(q.x, px0, t0)); z01 = fmaxf(z01, fmaf(q.x, px0, t1));
      z10 = fmaxf(z10, fmaf(q.x, px1, t0)); z11 = fmaxf(z11, fmaf(q.x, px1, t1));
    }
  }
}

// poses_ext != nullptr: test/profiling hook rendering explicit poses (srl_render_heightmap)
extern "C" __global__ void __launch_bounds__(SRL_RENDER_THREADS)
srl_k_render(DevParams P, uint8_t* __restrict__ obs_map, uint8_t* __restrict__ obs_obj, float* __restrict__ reward,
             uint8_t* __restrict__ done, const float* __restrict__ poses_ext, const int32_t* __restrict__ mesh_ext,
             const int32_t* __restrict__ nb_ext, float* __restrict__ height_ext) {
  extern __shared__ float4 lds_raw[];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int res = P.c.overhead_res, npx = res * res;
  RenderLds L;
  L.tile = (float*)lds_raw;
  L.planes = (float4*)(L.tile + npx);
  L.sx = (float*)(L.planes + SRL_PLANE_CAP);
  L.sR = L.sx + 3 * SRL_MAX_BODIES;
  L.mhdr = (int*)(L.sR + 9 * SRL_MAX_BODIES);
  L.bbox = (uint32_t*)(L.mhdr + 4 * SRL_MAX_BODIES);
  L.prange = (int*)(L.bbox + 4 * SRL_MAX_BODIES);
  L.reg = L.prange + 4 * SRL_MAX_BODIES;
  L.rowmask = (uint32_t*)(L.reg + 4 * SRL_MAX_BODIES);
  L.pi = (float*)L.planes;
  L.pu = L.pi + SRL_RENDER_THREADS;
  EnvHdr* h = &P.hdr[e];
  const float* gb = P.blob + (size_t)e * P.BLOB;
  const bool ext = poses_ext != nullptr;
#ifdef SRL_STAMPS
  long long _t0 = wall_clock64();
#endif
  const int nb = ext ? nb_ext[e] : h->nb;
  // header fields and the object map are requested up front: their latency overlaps the ray cast
  int g0 = 0, g1 = 0, g2 = 0, g3 = 0, pending = -1, mode = 0, hdone = 0;
  float prev_metric = 0.0f;
  if (!ext) {
    g0 = h->goal[0]; g1 = h->goal[1]; g2 = h->goal[2]; g3 = h->goal[3]; pending = h->pending;
    mode = h->mode; hdone = h->done; prev_metric = h->prev_metric;
  }
  const int rr = P.c.object_res * P.c.object_res;
  float om_pref[2] = {0.0f, 0.0f};
  if (!ext && pending >= 0) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + k * SRL_RENDER_THREADS < rr) om_pref[k] = P.objmap[(size_t)pending * rr + tid + k * SRL_RENDER_THREADS];
  }

  {
    float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4* t4 = (float4*)L.tile;
    for (int k = tid; k < npx / 4; k += SRL_RENDER_THREADS) t4[k] = z4;
  }
  if (tid < 8) L.rowmask[tid] = 0u;
  if (tid < nb) {
    q4 q; v3 x; int m;
    if (ext) {
      const float* p = poses_ext + ((size_t)e * SRL_MAX_BODIES + tid) * 7;
      x = V(p[0], p[1], p[2]); q.x = p[3]; q.y = p[4]; q.z = p[5]; q.w = p[6];
      m = mesh_ext[(size_t)e * SRL_MAX_BODIES + tid];
    } else {
      x = ld3(gb + P.OFF_X + 4 * tid);
      const float* qq = gb + P.OFF_Q + 4 * tid;
      q.x = qq[0]; q.y = qq[1]; q.z = qq[2]; q.w = qq[3];
      m = ((const int*)gb)[P.OFF_MESH + tid];
    }
    m3 R = quat_to_mat(q);
    st3(L.sx + 3 * tid, x);
#pragma unroll
    for (int i = 0; i < 9; ++i) L.sR[9 * tid + i] = R.m[i];
    const MeshHdr mh = P.mh[m];
    L.mhdr[4 * tid + 0] = mh.vo; L.mhdr[4 * tid + 1] = mh.nv; L.mhdr[4 * tid + 2] = mh.to; L.mhdr[4 * tid + 3] = mh.nt;
    L.bbox[4 * tid + 0] = f2o(1e30f); L.bbox[4 * tid + 1] = f2o(-1e30f);
    L.bbox[4 * tid + 2] = f2o(1e30f); L.bbox[4 * tid + 3] = f2o(-1e30f);
  }
  __syncthreads();
  RSTAMP(0);
  // ---- groups of rocks whose planes fit the staging area; the first group's face planes are requested
  //      now so that their latency overlaps the vertex pass
  int bs = 0, be = 0;
  {
    int np_group = 0;
    while (be < nb && np_group + L.mhdr[4 * be + 3] <= SRL_PLANE_CAP) { np_group += L.mhdr[4 * be + 3]; ++be; }
  }
  float4 myp = make_float4(0.0f, 0.0f, 0.0f, 0.0f); int myb = -1;
  {
    int b = bs, ts = 0, ntb = nb > 0 ? L.mhdr[3] : 0;
    while (b < be && tid >= ts + ntb) { ts += ntb; ++b; ntb = b < be ? L.mhdr[4 * b + 3] : 0; }
    if (b < be) { myp = P.mp[L.mhdr[4 * b + 2] + (tid - ts)]; myb = b; }
    if (tid < be) {   // region cursors of group 0
      int base = 0;
      for (int k = 0; k < tid; ++k) base += L.mhdr[4 * k + 3];
      L.reg[4 * tid + 0] = base; L.reg[4 * tid + 1] = base; L.reg[4 * tid + 2] = base + L.mhdr[4 * tid + 3];
    }
  }
  // ---- xy bounds of every rock: lanes over the flattened (rock, vertex) list
  {
    int b = 0, vs = 0, nvb = nb > 0 ? L.mhdr[1] : 0;
    for (int it = tid;; it += SRL_RENDER_THREADS) {
      while (b < nb && it >= vs + nvb) { vs += nvb; ++b; nvb = b < nb ? L.mhdr[4 * b + 1] : 0; }
      if (b >= nb) break;
      m3 R = ldm(L.sR + 9 * b);
      float4 lv = P.mv[L.mhdr[4 * b + 0] + (it - vs)];
      v3 a = mmul_add(R, V(lv.x, lv.y, lv.z), ld3(L.sx + 3 * b));
      atomicMin(&L.bbox[4 * b + 0], f2o(a.x)); atomicMax(&L.bbox[4 * b + 1], f2o(a.x));
      atomicMin(&L.bbox[4 * b + 2], f2o(a.y)); atomicMax(&L.bbox[4 * b + 3], f2o(a.y));
    }
  }
  __syncthreads();
  RSTAMP(1);
  if (tid < nb) {
    int i0 = 0, i1 = -1, j0 = 0, j1 = -1;
    bool okx = pixel_range(o2f(L.bbox[4 * tid + 0]), o2f(L.bbox[4 * tid + 1]), P.inv_px, res, i0, i1);
    bool oky = pixel_range(o2f(L.bbox[4 * tid + 2]), o2f(L.bbox[4 * tid + 3]), P.inv_px, res, j0, j1);
    int w2 = 0, items = 0;
    if (okx && oky) { w2 = (j1 - j0 + 2) >> 1; items = ((i1 - i0 + 2) >> 1) * w2; }   // 2 x 2 pixel quads
    L.prange[4 * tid + 0] = i0 | (i1 << 16); L.prange[4 * tid + 1] = j0 | (j1 << 16); L.prange[4 * tid + 2] = w2; L.prange[4 * tid + 3] = items;
    if (items > 0)
      for (int w = i0 >> 5; w <= i1 >> 5; ++w) {
        const int lo = max(i0 - 32 * w, 0), hi = min(i1 - 32 * w, 31);
        atomicOr(&L.rowmask[w], (0xffffffffu >> (31 - hi)) & (0xffffffffu << lo));
      }
  }
  // epilogue constants (the rows no rock reaches are written out before the ray cast, see below)
  const float nearp = SRL_FAR - P.c.max_z;
  const float den = fmaxf(P.c.max_z, P.c.object_max_dimension);   // env.py:171-172
  const float gz = P.goal_z;
  const uint32_t gbyte = (uint8_t)((gz * 255.0f) / den);
  const uint32_t zbyte = (uint8_t)((0.0f * 255.0f) / den);
  const float h_empty = elev_overhead(P, depth_encode(SRL_FAR - 0.0f, nearp, SRL_FAR));
  const uint32_t b_empty = (uint8_t)((h_empty * 255.0f) / den);
  float* Hout = ext ? height_ext + (size_t)e * npx : P.H + (size_t)e * npx;
  uint8_t* om = ext ? nullptr : obs_map + (size_t)e * npx * 2;
  __syncthreads();
  // ---- rows no rock reaches hold the empty-pixel constants: their H / observation bytes leave now, so
  //      that HBM drains them while the ray cast computes.  cov bit k: this thread's k-th pixel group lies
  //      in a row some rock may reach (handled by the epilogue).  The IoU sums of all groups are taken in
  //      the epilogue, in the fixed order.
  uint32_t cov = 0u;
  {
    int k = 0;
    const float4 he4 = make_float4(h_empty, h_empty, h_empty, h_empty);
    for (int g = tid; g < npx / 4; g += SRL_RENDER_THREADS, ++k) {
      const int k0 = g * 4, i = k0 / res, jb = k0 - i * res;
      if ((L.rowmask[i >> 5] >> (i & 31)) & 1u) { cov |= 1u << k; continue; }
      ((float4*)Hout)[g] = he4;
      if (om) {
        const bool row_in = (i >= g0 && i < g0 + g2);
        uint32_t pk[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int j = jb + t;
          pk[t] = b_empty | (((row_in && (j >= g1 && j < g1 + g3)) ? gbyte : zbyte) << 8);
        }
        ((uint2*)om)[g] = make_uint2(pk[0] | (pk[1] << 16), pk[2] | (pk[3] << 16));
      }
    }
  }
  RSTAMP(6);
  while (bs < nb) {
    if (bs > 0) {   // later groups: cursors, then request the planes
      if (tid >= bs && tid < be) {
        int base = 0;
        for (int k = bs; k < tid; ++k) base += L.mhdr[4 * k + 3];
        L.reg[4 * tid + 0] = base; L.reg[4 * tid + 1] = base; L.reg[4 * tid + 2] = base + L.mhdr[4 * tid + 3];
      }
      int b = bs, ts = 0, ntb = L.mhdr[4 * bs + 3];
      myb = -1;
      while (b < be && tid >= ts + ntb) { ts += ntb; ++b; ntb = b < be ? L.mhdr[4 * b + 3] : 0; }
      if (b < be) { myp = P.mp[L.mhdr[4 * b + 2] + (tid - ts)]; myb = b; }
      __syncthreads();
    }
    // (a) world-frame plane; slot by wave-aggregated cursor bumps (up from the front, down from the back;
    //     the order inside a region is irrelevant: min / max)
    {
      int key = -1;
      if (myb >= 0) {
        myp = make_rplane(myp, ldm(L.sR + 9 * myb), ld3(L.sx + 3 * myb));
        key = (myb << 1) | __float_as_int(myp.w);
      }
      unsigned long long todo = __ballot(key >= 0);
      int slot = -1;
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader);
        const unsigned long long m = __ballot(key == k);
        const int cnt = __popcll(m);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        int basev = 0;
        if (lane == leader) basev = (k & 1) ? atomicSub(&L.reg[4 * (k >> 1) + 2], cnt) : atomicAdd(&L.reg[4 * (k >> 1) + 1], cnt);
        basev = __shfl(basev, leader);
        if (key == k) slot = (k & 1) ? (basev - 1 - rank) : (basev + rank);
        todo &= ~m;
      }
      if (slot >= 0) L.planes[slot] = myp;
    }
    __syncthreads();
  RSTAMP(2);
    // (b) lanes over the flattened (rock, 2 x 2 pixel quad) list; when the list is short each quad is
    //     shared by S adjacent lanes that split the planes and combine with shuffles
    {
      int total = 0;
      for (int b = bs; b < be; ++b) total += L.prange[4 * b + 3];
      int lg = 0;
      while (lg < 3 && (total << (lg + 1)) <= SRL_RENDER_THREADS) ++lg;
      const int S = 1 << lg, s = tid & (S - 1);
      int b = bs, is = 0, nib = L.prange[4 * bs + 3];
      for (int it = tid >> lg;; it += SRL_RENDER_THREADS >> lg) {
        while (b < be && it >= is + nib) { is += nib; ++b; nib = b < be ? L.prange[4 * b + 3] : 0; }
        if (b >= be) break;
        const int w2 = L.prange[4 * b + 2], jj = L.prange[4 * b + 1], ii = L.prange[4 * b + 0];
        const int j0 = jj & 0xffff, j1 = jj >> 16, i1 = ii >> 16;
        const int p = it - is;
        const int di = p / w2;
        const int i = (ii & 0xffff) + 2 * di, j = j0 + 2 * (p - di * w2);
        const bool row2 = i + 1 <= i1, col2 = j + 1 <= j1;
        const float px0 = ((float)i + 0.5f) * P.px, px1 = ((float)(i + 1) + 0.5f) * P.px;
        const float py0 = ((float)j + 0.5f) * P.px, py1 = ((float)(j + 1) + 0.5f) * P.px;
        const int base = L.reg[4 * b + 0], nup = L.reg[4 * b + 1] - base;
        const int dn0 = L.reg[4 * b + 2], ndn = base + L.mhdr[4 * b + 3] - dn0;
        float h00 = 1e30f, h01 = 1e30f, h10 = 1e30f, h11 = 1e30f;
        float l00 = -1e30f, l01 = -1e30f, l10 = -1e30f, l11 = -1e30f;
        plane_sweep<true>(L.planes + base, nup, s, S, px0, px1, py0, py1, h00, h01, h10, h11);
        plane_sweep<false>(L.planes + dn0, ndn, s, S, px0, px1, py0, py1, l00, l01, l10, l11);
        for (int m = 1; m < S; m <<= 1) {
          h00 = fminf(h00, __shfl_xor(h00, m)); h01 = fminf(h01, __shfl_xor(h01, m));
          h10 = fminf(h10, __shfl_xor(h10, m)); h11 = fminf(h11, __shfl_xor(h11, m));
          l00 = fmaxf(l00, __shfl_xor(l00, m)); l01 = fmaxf(l01, __shfl_xor(l01, m));
          l10 = fmaxf(l10, __shfl_xor(l10, m)); l11 = fmaxf(l11, __shfl_xor(l11, m));
        }
        if (s == 0) {
          int* t0p = (int*)&L.tile[i * res + j];   // positive floats order as ints
          if (l00 <= h00 && h00 > 0.0f) atomicMax(t0p, __float_as_int(h00));
          if (col2 && l01 <= h01 && h01 > 0.0f) atomicMax(t0p + 1, __float_as_int(h01));
          if (row2 && l10 <= h10 && h10 > 0.0f) atomicMax(t0p + res, __float_as_int(h10));
          if (row2 && col2 && l11 <= h11 && h11 > 0.0f) atomicMax(t0p + res + 1, __float_as_int(h11));
        }
      }
    }
    __syncthreads();
  RSTAMP(3);
    bs = be;
    {
      int np_group = 0;
      while (be < nb && np_group + L.mhdr[4 * be + 3] <= SRL_PLANE_CAP) { np_group += L.mhdr[4 * be + 3]; ++be; }
    }
  }
  __syncthreads();
  RSTAMP(4);
  // ---- epilogue: depth codec, H out, uint8 pack, IoU partial sums (4 pixels per thread per round)
  float spi = 0.0f, spu = 0.0f;
  {
    int k = 0;
    for (int g = tid; g < npx / 4; g += SRL_RENDER_THREADS, ++k) {
      const int k0 = g * 4;
      const int i = k0 / res, jb = k0 - i * res;
      const bool row_in = (i >= g0 && i < g0 + g2);
      if (!((cov >> k) & 1u)) {   // written out above; sums only
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int j = jb + t;
          if (row_in && (j >= g1 && j < g1 + g3)) { spi += fminf(h_empty, gz); spu += fmaxf(h_empty, gz); }
          else spu += fmaxf(h_empty, 0.0f);
        }
        continue;
      }
      const float4 z4 = ((const float4*)L.tile)[g];
      const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
      const bool any = (z4.x > 0.0f) || (z4.y > 0.0f) || (z4.z > 0.0f) || (z4.w > 0.0f);
      float hv[4];
      uint32_t pk[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float hh = h_empty;
        uint32_t hb = b_empty;
        if (any && zz[t] > 0.0f) {
          hh = elev_overhead(P, depth_encode(SRL_FAR - zz[t], nearp, SRL_FAR));
          hb = (uint8_t)((hh * 255.0f) / den);
        }
        hv[t] = hh;
        const int j = jb + t;
        const bool in = row_in && (j >= g1 && j < g1 + g3);
        if (in) { spi += fminf(hh, gz); spu += fmaxf(hh, gz); }
        else spu += fmaxf(hh, 0.0f);
        pk[t] = hb | ((in ? gbyte : zbyte) << 8);
      }
      ((float4*)Hout)[g] = make_float4(hv[0], hv[1], hv[2], hv[3]);
      if (om) ((uint2*)om)[g] = make_uint2(pk[0] | (pk[1] << 16), pk[2] | (pk[3] << 16));
    }
  }
  RSTAMP(5);
  if (ext) return;
  // ---- object observation (O2 from the per-mesh cache, requested at kernel start; empty map when nothing is pending)
  {
    uint8_t* oo = obs_obj + (size_t)e * rr;
    const float empty = elev_object(P, 1.0f);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + k * SRL_RENDER_THREADS;
      if (idx < rr) oo[idx] = (uint8_t)(((pending >= 0 ? om_pref[k] : empty) * 255.0f) / den);
    }
    for (int idx = tid + 2 * SRL_RENDER_THREADS; idx < rr; idx += SRL_RENDER_THREADS)
      oo[idx] = (uint8_t)(((pending >= 0 ? P.objmap[(size_t)pending * rr + idx] : empty) * 255.0f) / den);
  }
  // ---- halving tree over the 512 partials: cross-wave stages through LDS, in-wave stages by shuffles
  L.pi[tid] = spi; L.pu[tid] = spu;
  __syncthreads();
  if (tid < 256) { L.pi[tid] += L.pi[tid + 256]; L.pu[tid] += L.pu[tid + 256]; }
  __syncthreads();
  if (tid < 128) { L.pi[tid] += L.pi[tid + 128]; L.pu[tid] += L.pu[tid + 128]; }
  __syncthreads();
  float sum_i = 0.0f, sum_u = 0.0f;
  if (tid < 64) {
    sum_i = L.pi[tid] + L.pi[tid + 64]; sum_u = L.pu[tid] + L.pu[tid + 64];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { sum_i = sum_i + __shfl_down(sum_i, s); sum_u = sum_u + __shfl_down(sum_u, s); }   // p[t] += p[t+s], t < s
  }
  // ---- K5: reward = scale * (metric_t - metric_{t-1})  (rewarder.py:176-179)
  if (tid == 0) {
    if (mode == 0) {
      float mv;
      if (P.c.metric == SRL_METRIC_IOU) mv = sum_i / sum_u;
      else if (P.c.metric == SRL_METRIC_OR) mv = sum_i / ((float)(g2 * g3) * gz);
      else mv = discounted_metric(P, h, gb);
      reward[e] = (mv - prev_metric) * P.scale;
      h->prev_metric = mv;
      done[e] = (uint8_t)hdone;
    } else {   // reset step (env.py:235-236) or rejected action
      reward[e] = 0.0f;
      done[e] = 0;
    }
  }
}

// K3: underside map of one mesh at the spawn pose.  One workgroup per mesh.
extern "C" __global__ void __launch_bounds__(256) srl_k_objmap(DevParams P, float* __restrict__ out) {
  __shared__ float4 planes[SRL_MAX_TRIS];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int r = P.c.object_res;
  const float half = P.c.object_max_dimension * 0.5f;
  const MeshHdr mh = P.mh[m];
  m3 I;
#pragma unroll
  for (int k = 0; k < 9; ++k) I.m[k] = (k % 4 == 0) ? 1.0f : 0.0f;
  const v3 xs = V(mh.cx + half, mh.cy + half, mh.cz);   // map coordinates: link frame shifted to start at 0
  for (int t = tid; t < mh.nt; t += 256) planes[t] = make_rplane(P.mp[mh.to + t], I, xs);
  __syncthreads();
  const float nearp = SRL_FAR - half, farp = SRL_FAR + half;
  for (int k = tid; k < r * r; k += 256) {
    int i = k / r, j = k - i * r;
    float px = ((float)i + 0.5f) * P.px, py = ((float)j + 0.5f) * P.px;
    float lo, hi;
    float z = 1e30f;
    if (ray_cast(planes, mh.nt, px, py, lo, hi)) z = lo;
    float d = z > 1e29f ? 1.0f : depth_encode(SRL_FAR + z, nearp, farp);
    out[(size_t)m * r * r + k] = elev_object(P, d);
  }
}
