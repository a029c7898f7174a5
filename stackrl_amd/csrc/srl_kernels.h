// srl_kernels.h — shared host/device parameter blocks of libstackrl_hip.
#pragma once
#include <stdint.h>
#include "../../include/srl_types.h"

// Per-mesh header in the device mesh table.
struct MeshHdr {
  int32_t vo, nv, to, nt;       // offsets/counts into mesh_verts (float4) / mesh_tris (uchar4)
  float inv_mass;
  float iix, iiy, iiz;          // inverse inertia diagonal (body frame): box inertia of the AABB
  float radius;                 // bounding radius about the centre of mass
  float cx, cy, cz;             // centre of mass in the URDF link frame
  int32_t eo, ne;               // offset/count into mesh_edges (uchar4: vertex a < vertex b, the two faces sharing it)
};

// Per-env scalar state (global memory).
struct EnvHdr {
  int32_t nb;                   // placed bodies
  int32_t done;                 // StackEnv._done (env.py:219-220)
  uint32_t episode;             // episode counter (RNG key part)
  int32_t list_pos;             // next index into ids (episode_list.pop, env.py:243-247); with ordering freedom the
                                // number of rocks still unplaced: ids[0 .. list_pos) (simulator.py:343-378)
  int32_t pending;              // mesh id waiting at the spawn pose, -1 = none (Simulator._new)
  int32_t mode;                 // what the last step did: 0 placement, 1 reset, 2 rejected action
  int32_t goal[4];              // u, v, h, w (rewarder.py:255-257)
  float prev_metric[4];         // Rewarder._memory (rewarder.py:100)
  int32_t substeps[2];          // Simulator.n_steps
  int32_t sweeps;               // solver sweeps of the last step, all its sub-steps together (telemetry)
  int32_t status;               // SRL_ST_* bits
  int32_t ncolour;              // contact-graph colours, -1 = recolour
  int32_t has_script;
  int32_t ids[SRL_MAX_BODIES];
  int32_t script_ids[SRL_MAX_BODIES];
  int32_t script_goal[4];
#ifdef SRL_STAMPS
  long long stamps[8];          // diagnostic build only: accumulated wall-clock ticks per sub-step phase
  long long stamps2[4];         // narrowphase of slot 0: refresh, gjk, insert ticks, calls
  long long rstamps[8];         // render kernel phases
  long long diag[6];            // sub-steps with a solve, of those without a pair point, colours summed, sweeps summed, pair turns per sweep summed, and what a level schedule of the same rows would need (tools/stamps.py)
  long long hwid[2];            // HW_ID | XCC_ID << 32 of the settle workgroup's two first waves (tools/diag_placement.py)
#endif
};

// Everything a kernel needs, passed by value.
struct DevParams {
  srl_config c;
  // derived scalars (computed once on the host in double, rounded to float like the oracle does)
  float px, inv_px, lin_damp, ang_damp, goal_z, scale, elev_num, obj_c1, obj_c2;
  int32_t max_substeps, goal_size, goal_min_h, goal_max_h, goal_min_w, goal_max_w, AW, A;
  int32_t n_mesh, VS /*vertex stride*/, NS /*manifold slots*/, NP /*body pairs*/;
  uint32_t seed, sample_counter;
  int32_t force_reset;
  // persistent blob layout (32-bit words)
  int32_t OFF_X, OFF_Q, OFF_V, OFF_W, OFF_PX, OFF_PQ, OFF_MESH, OFF_GM, OFF_MAN, OFF_SOP, OFF_POS, OFF_COL, BLOB;
  // scratch layout (words, after the blob in LDS)
  int32_t S_R, S_IW, S_AMIN, S_AMAX, S_BC, S_WV, S_LV, S_USED, S_MISC, S_PAIR, LDS_WORDS;
  // device pointers
  EnvHdr* hdr;
  float* blob;            // [n_envs][BLOB]
  float* H;               // [n_envs][res*res]
  const MeshHdr* mh;
  const float4* mv;       // mesh vertices, COM frame
  const uchar4* mt;       // triangles
  const uchar4* me;       // edges: (va, vb, fa, fb), closed triangulated surface: every edge has exactly two faces
  const float4* mp;       // face planes, COM frame: unit normal xyz, offset d
  const float* objmap;    // [n_mesh][n_orient][ores*ores] underside maps (O2), one per observable orientation
  // observable orientations of the pending rock (TestStackEnv, observer.py:127-140): quaternion i = inverse of the yaw
  // i * 2 pi / n_orient ("orientation of the object relative to the view"); n_orient = 1: identity only (Stack-v0)
  int32_t n_orient;
  int32_t n_slots;        // object maps per observation: n_orient, or episode_length * n_orient with ordering freedom
  float orient_q[SRL_MAX_ORIENT][4];
  // depth codec of the overhead camera as a table (srl_k_codec_table): the ray-cast height z enters the codec only as
  // t = fl(FAR - z), a float32 in [512, 1024) and therefore on a lattice of 2^-14 — codec[v], v = 1 .. codec_n, =
  // (elevation bits, observation byte) for t = near + (codec_n - v) 2^-14, computed on the device by the codec's own
  // expressions (observer.py:259-260, env.py:171-172), so a lookup returns the bits the arithmetic would; codec[0]
  // holds the constants of a pixel that saw no rock.  Rows grow with z, so the render tile keeps the max row per pixel.
  // srl_create requires FAR - max_z >= 512 and max_z <= 4 (65,536 rows).
  const uint2* codec;
  int32_t codec_n;
  // constants of the render epilogue, evaluated once on the device by the expressions the oracle uses (rows 0, n + 1 and
  // n + 2 of the codec table) and read back by srl_create: elevation / observation byte of a pixel that saw no rock,
  // goal and zero bytes of the goal channel (env.py:171-172), byte of an empty object-map pixel
  float h_empty;
  uint32_t b_empty, gbyte, zbyte, obj_empty_byte;
  // walk of a thread over its pixel groups (group g = tid + 512 k = row i, columns jb .. jb + 3): per round +walk_di
  // rows and +walk_dj columns; res_magic = floor(2^32 / res) + 1 (exact quotients of x < 2^32 / res by a multiply-high)
  int32_t walk_di, walk_dj;
  uint32_t res_magic;
  const uint8_t* objmap_u8;   // [n_mesh][n_orient][ores*ores] the same maps as observation bytes (env.py:171-172)
  int32_t* flags;         // [1] accumulated error bits since the last srl_sync_status
};
