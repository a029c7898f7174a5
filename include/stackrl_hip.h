/* stackrl_hip.h — C-ABI of libstackrl_hip.so, the MI355X-native batched Stack-v0 env step.
 *
 * This is the drop-in boundary for the reference's `stackrl.envs.make('Stack-v0', n_parallel=B)`
 * object (stackrl/envs/utils.py:185-300 `Env`, :302-576 `ParallelEnv`): one handle owns B
 * independent envs on one GPU; `srl_step` replaces the `conn.send((STEP, a))` / `conn.recv()`
 * fan-out (utils.py:468-486, :540-543, :554-566) plus everything each worker process runs for it
 * (`StackEnv.step` env.py:233-264 -> `Observer.pose` observer.py:392-421 -> `Simulator.step`
 * simulator.py:190-258 -> `Observer.__call__` observer.py:249-277 -> `Rewarder.__call__`
 * rewarder.py:144-179 -> `StackEnv.observation` env.py:225-231).
 *
 * Conventions
 *  - plain C, no torch / HIP types in the signatures; `stream` is a hipStream_t passed as void*
 *    (NULL = the null stream).
 *  - "dev" pointers are device memory owned by the caller; "host" pointers are host memory.
 *  - every function returns an SRL_* code (include/srl_types.h); `srl_last_error()` gives the
 *    message of the calling thread's last failure.
 *  - `srl_reset/srl_step/srl_sample` only enqueue work on `stream` and return; the caller's
 *    buffers are valid when the stream reaches that point.  No device allocation, free or blocking
 *    copy happens in them in steady state (the first launch after srl_create / srl_load_meshes /
 *    srl_seed uploads the parameter block with one blocking copy; with srl_set_profiling on, HIP
 *    events are created on demand).
 *  - a handle is not thread-safe; distinct handles are independent: they share no device or host
 *    state (no process-wide tables, no common scratch buffers), so handles on different streams may
 *    step concurrently (`tests/test_parity_gpu.py::test_two_handles_on_two_streams_*`).
 */
#ifndef STACKRL_HIP_H_
#define STACKRL_HIP_H_

#include "srl_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srl_env srl_env;

/* Stack-v0 defaults: env.py:28-51 under the registry kwargs of envs/stack/__init__.py:4-8. */
int srl_config_default(srl_config* cfg);

/* `envs.make(...)` / `ParallelEnv.__init__` + `start` (utils.py:311-448): allocates all device state
 * for cfg->n_envs envs on the current HIP device. */
int srl_create(const srl_config* cfg, srl_env** out);
/* `ParallelEnv.terminate` (utils.py:450-466). */
void srl_destroy(srl_env* env);
const char* srl_last_error(void);

/* Mesh pool = the URDF list `data.generated(name='[5-9]?')` (envs/data/__init__.py:39-83) preloaded
 * once instead of `loadURDF` per step (simulator.py:297-308).  Host pointers, OBJ/URDF link frame:
 * verts float[vert_off[n_mesh]][3]; tris int32[tri_off[n_mesh]][3] per-mesh local, outward CCW;
 * mass_com float[n_mesh][4] = mass, inertial origin xyz (template.urdf:8-9). */
int srl_load_meshes(srl_env* env, const float* verts, const int32_t* vert_off, const int32_t* tris,
                    const int32_t* tri_off, const float* mass_com, int32_t n_mesh);

/* `ParallelEnv.seed` (utils.py:522-532) / `StackEnv.seed` (env.py:341-346): env i uses
 * seed + env_index_offset + i (mod 2^32) as the key of its counter RNG; episode counters restart. */
int srl_seed(srl_env* env, uint32_t seed);

/* Explicit episode script for the NEXT reset of every env (host pointers): mesh_ids int32[n][L],
 * goal_rect int32[n][4] = (u, v, h, w).  Stands in for the reference's RNG streams
 * (env.py:268-272, rewarder.py:211-259), which are gym-version dependent. */
int srl_set_script(srl_env* env, const int32_t* mesh_ids, const int32_t* goal_rect);

/* `ParallelEnv.reset` (utils.py:488-503, :545-552): obs_map u8[n][H][W][2], obs_obj u8[n][h][w][1].
 * With orientation_freedom = k > 0 (`TestStackEnv`, env.py:443-470) obs_obj is u8[n][2^k][h][w][1]: one object map per
 * observable orientation (the overhead map is returned once, not 2^k copies as env.py:472-480 stacks them).
 * With ordering_freedom = 1 (`TestSimulator`, simulator.py:343-378) obs_obj is u8[n][L * 2^k][h][w][1]: the maps of the
 * rocks still unplaced first (rock-major, orientation-minor, observer.py:310-327), empty maps after them — the fixed-size
 * form of the reference's shrinking list (env.py:596-608). */
int srl_reset(srl_env* env, void* obs_map_dev, void* obs_obj_dev, void* stream);

/* `ParallelEnv.step` (utils.py:468-486): action int64[n]; reward float[n] (float[n][4] with metric 'all' = IoU, OR, DIoU,
 * DOR; float[n][2] with 'eval' = IoU, AD: the dict of rewarder.py:147-158 as columns); done uint8[n].
 * Auto-reset semantics of env.py:235-236 are kept: a step on a finished env returns the reset
 * observation, reward 0, done 0.  With orientation_freedom > 0 the action is orientation * A + pixel
 * (the reference's `(index, action)` tuple, env.py:485-494) and the rock is placed in that orientation.  With
 * ordering_freedom the index also names the rock: action = (rock * 2^k + orientation) * A + pixel, rock counted among
 * those still unplaced (it is popped from the list, simulator.py:372-378); an index past the maps on show is an invalid
 * action (env.py:484); the episode ends when no rock is left (env.py:513-514). */
int srl_step(srl_env* env, const int64_t* action_dev, void* obs_map_dev, void* obs_obj_dev,
             float* reward_dev, uint8_t* done_dev, void* stream);

/* `ParallelEnv.sample` (utils.py:534-538): uniform actions in [0, A), int64[n] on the device ([0, maps on show * A) for
 * `TestStackEnv`). */
int srl_sample(srl_env* env, int64_t* action_dev, void* stream);

/* Blocks until `stream` is idle and reports what the reference would have raised during the steps
 * since the last call: SRL_EINVAL_ACTION ("Invalid action.", env.py:238), SRL_ESIM_DIVERGED
 * (simulator.py:221-224 / :242-245), else SRL_OK. */
int srl_sync_status(srl_env* env, void* stream);

/* Telemetry (host pointers, synchronises the device; any pointer may be NULL):
 * `Simulator.poses` (simulator.py:90-93) as float[n][SRL_MAX_BODIES][8] = pos xyz, quat xyzw, mesh id;
 * n_bodies int32[n]; `Simulator.n_steps` (simulator.py:79-83) int32[n][2]; status bits int32[n]. */
int srl_get_state(srl_env* env, float* poses, int32_t* n_bodies, int32_t* substeps, int32_t* status);
/* `Observer.state` (observer.py:365-368) float[n][H*W], float[n][2^orientation_freedom][h*w]; `Rewarder` goal rect int32[n][4]. */
int srl_get_maps(srl_env* env, float* height, float* object_map, int32_t* goal_rect);
/* State injection for closed-form physics tests (host pointers, synchronises; either pointer may be NULL = leave as is):
 * `pb.resetBasePositionAndOrientation` (simulator.py:313) and `pb.resetBaseVelocity` (simulator.py:214) for every placed
 * body of every env, in the layouts srl_get_state / srl_get_velocities return (the mesh-id column is ignored). */
int srl_set_body_state(srl_env* env, const float* poses, const float* velocities);
/* `pb.stepSimulation` (simulator.py:219, :240, :320) x n_substeps on every env: no placement, no stop criterion, no
 * render — the raw sub-step the three loops of `Simulator.step` are built from. */
int srl_step_simulation(srl_env* env, int32_t n_substeps, void* stream);
/* solver telemetry int32[n]: sequential-impulse sweeps run by the last step of each env, all its sub-steps together
 * (each sub-step runs at most solver_iterations sweeps and ends them early on the residual threshold) */
int srl_get_sweeps(srl_env* env, int32_t* sweeps);
/* velocities float[n][SRL_MAX_BODIES][8] = lin xyz 0, ang xyz 0 */
int srl_get_velocities(srl_env* env, float* vel);
/* contact telemetry: deepest penetration (m) and number of manifold points per env */
int srl_get_contacts(srl_env* env, float* max_penetration, int32_t* n_points);

/* Renderer on explicit poses (test / profiling hook for the O1 row, observer.py:252-260):
 * poses_dev float[n][SRL_MAX_BODIES][7] (COM frame), mesh_ids_dev int32[n][SRL_MAX_BODIES],
 * n_bodies_dev int32[n] -> height_dev float[n][H*W]. */
int srl_render_heightmap(srl_env* env, const float* poses_dev, const int32_t* mesh_ids_dev,
                         const int32_t* n_bodies_dev, float* height_dev, void* stream);
/* O2 row: underside map(s) of one mesh (observer.py:262-277), host output float[2^orientation_freedom][h*w]. */
int srl_get_object_map(srl_env* env, int32_t mesh_id, float* object_map);

/* Per-kernel HIP-event timing (bench.py's roofline leg).  enable != 0 brackets every kernel launch of
 * srl_reset/srl_step with events on the launch stream; srl_get_kernel_times synchronises and returns the
 * accumulated milliseconds and launch counts since the last call: index 0 = settle (K1+K4),
 * 1 = render (K2+K5+obs pack), 2 = the staging kernel: the rocks' render records as a kernel of its own, for the explicit-pose hook and
 * for a step after a test hook moved bodies; in steady state the records are made in the settle kernel's tail, csrc/stage.h. */
int srl_set_profiling(srl_env* env, int32_t enable);

/* Tuning hint, between srl_create and srl_load_meshes: the number of envs that step on this device at the same time over
 * ALL handles of the caller (0 = this handle alone).  The reference's `ParallelEnv` has one process per env and a caller
 * is free to hold its batch as several handles (independent shards that step as their actions arrive, utils.py:468-486);
 * the settle kernel comes in a latency-oriented and a throughput-oriented build (9 - 16 rocks) and this number, not the
 * handle's own n_envs, says which one fits.  Results do not depend on it (the parity tests run both builds). */
int srl_set_concurrent_envs(srl_env* env, int32_t n_envs_on_device);

/* Launch order of the settle kernel's workgroups (one per env).  The reference's `ParallelEnv` collects its worker
 * processes' results as they come (utils.py:540-543) and a vectorised step lasts as long as its slowest env (the stop
 * criterion of simulator.py:322-335); a batch that outnumbers the workgroups the device holds at once therefore starts the
 * envs with the longest expected settle first — those whose rock `Observer.pose` (observer.py:405-413) will release highest,
 * evaluated and sorted on the device before the step kernel.  mode -1 (default): by batch size (>= 2,048 envs), 0: index
 * order, 1: always ordered (at most 16,384 envs per handle).  Envs are independent: results do not depend on the order
 * (the parity tests run both). */
int srl_set_launch_order(srl_env* env, int32_t mode);
int srl_get_kernel_times(srl_env* env, float* ms3, int32_t* launches3);
/* the two kernels of an ordered launch (keys + sort), which run before the settle kernel and are not part of its time:
 * accumulated milliseconds and launches since the last call; call srl_get_kernel_times first. */
int srl_get_order_kernel_times(srl_env* env, float* ms, int32_t* launches);
/* Test hook: keys [n_envs] and permutation [n_envs] of the latest ordered launch (order[k] = the env workgroup k served,
 * ascending keys = highest release first).  Synchronises the device. */
int srl_get_launch_order(srl_env* env, unsigned long long* keys, int32_t* order);

/* Test hook: the staged rock records of the latest step (csrc/stage.h: the settle kernel's tail -> srl_k_render), host array
 * float[n_envs][episode_length][srl_stage_record_stride()][4]: per rock its pixel bounding box, up-facing planes, outline
 * sides and, per row of ray-cast items, the columns and the ranges of the two lists that row sweeps (layout: render.hip).
 * Synchronises the device. */
int srl_get_stage_records(srl_env* env, float* records, int64_t n_floats);
int32_t srl_stage_record_stride(void);

/* How this library was built: "SRL_BUILD_INFO<variant|hash>" — variant "vectorised+rewritten" (clang's SLP vectoriser on and
 * the pass of stackrl_amd/isa_fix.py over the compiled assembly) or "safe" (built in one go without the vectoriser), hash =
 * sha256 prefix of the sources and flags (stackrl_amd/build.py; bench.py prints both). */
const char* srl_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* STACKRL_HIP_H_ */
