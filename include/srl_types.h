/* srl_types.h — plain-C data types shared by the C-ABI (include/stackrl_hip.h)
 * and by the CPU oracle (oracle/srl_oracle.h).  Types only: no code lives here,
 * so the product library and the test oracle share a vocabulary but not an
 * implementation.
 *
 * Every field cites the reference parameter it carries (paths relative to
 * menezesandre/stackrl).
 */
#ifndef SRL_TYPES_H_
#define SRL_TYPES_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Hard limits of the device layout. */
#define SRL_MAX_BODIES 32      /* episode_length <= 32 (BASELINE configs: 8/16/32) */
#define SRL_MAX_VERTS 128      /* reference pool: 24..70 vertices per rock        */
#define SRL_MAX_TRIS 252       /* reference pool: 44..136 triangles per rock      */
#define SRL_ACTION_HOLD (-2)  /* action value: the env sits this call out (no placement, no auto-reset; observation
                                 unchanged, reward 0, done 0) — lets the host run the start placements of the envs that
                                 were just reset (`StartedStackEnv.reset`, env.py:436-440) while the others wait */
#define SRL_MAX_ORIENT 16      /* 2^orientation_freedom <= 16                     */

/* Reward metrics, rewarder.py:7-14. */
/* 'all' (rewarder.py:157-158): the four metrics' rewards at once, reward is float[n][4] in this order; 'eval'
 * (rewarder.py:147-156): float[n][2] = the IoU reward and 'AD', the change of the average discount of all rocks
 * (not scaled). */
enum { SRL_METRIC_IOU = 0, SRL_METRIC_OR = 1, SRL_METRIC_DIOU = 2, SRL_METRIC_DOR = 3, SRL_METRIC_ALL = 4, SRL_METRIC_EVAL = 5 };

/* Return codes (the Python shim maps them to the reference's exception types:
 * AssertionError env.py:238, RuntimeError simulator.py:221-224, ValueError env.py:169). */
enum {
  SRL_OK = 0,
  SRL_EINVAL = 1,          /* bad argument / configuration                        */
  SRL_EINVAL_ACTION = 2,   /* action outside [0, A) — env.py:238                  */
  SRL_ESIM_DIVERGED = 3,   /* max sub-steps reached in some env — simulator.py:221 */
  SRL_EHIP = 4,            /* HIP runtime error (message in srl_last_error)       */
  SRL_ENOMESH = 5          /* srl_load_meshes not called / pool smaller than L    */
};

/* Per-env status bits reported by srl_get_status. */
enum {
  SRL_ST_DIVERGED = 1,      /* sub-step cap hit (reference raises RuntimeError)  */
  SRL_ST_PAIR_OVERFLOW = 2, /* more close pairs than manifold slots              */
  SRL_ST_BAD_ACTION = 4     /* action out of range: step was not applied         */
};

typedef struct srl_config {
  /* --- batching / sharding (utils.py:424-448) --- */
  int32_t n_envs;            /* envs owned by this handle                         */
  int32_t env_index_offset;  /* global index of env 0 (rank*B/G); seed_i = seed + offset + i (utils.py:433) */
  /* --- StackEnv ctor, env.py:28-51 (Stack-v0 values in comments) --- */
  int32_t episode_length;    /* 30; BASELINE configs 8/16/32                      */
  int32_t overhead_res;      /* H = W = 2^resolution_factor * observable_size_ratio = 128 */
  int32_t object_res;        /* h = w = 2^resolution_factor = 32                  */
  float object_max_dimension;/* 0.125                                             */
  float max_z;               /* 0.375                                             */
  float sim_time_step;       /* 0.01                                              */
  float gravity;             /* 9.8                                               */
  float velocity_threshold;  /* 0.01                                              */
  int32_t smooth_placing;    /* 1                                                 */
  int32_t max_substeps;      /* int(MAX_STEP_TIME/time_step), simulator.py:46; 0 = derive */
  /* --- Rewarder, rewarder.py:17-27 --- */
  int32_t metric;            /* SRL_METRIC_*; None -> IoU (rewarder.py:113-114); ALL / EVAL: several rewards per env */
  float goal_size_ratio;     /* 0.25 (scalar-area branch rewarder.py:225-237)     */
  float reward_scale;        /* 1.0; <= 0 means None -> n_objects (rewarder.py:97) */
  int32_t reward_pexp;       /* reward_params: integer exponent, 2; < 0 = None    */
  int32_t reward_oexp;       /* idem for rotation; 2                               */
  /* --- solver definition: the values pybullet's physics server runs with when the client only calls setTimeStep
   *     (simulator.py:143; DESIGN.md section 5 lists every value with its source) --- */
  int32_t solver_iterations; /* cap on sequential-impulse sweeps per sub-step: pybullet numSolverIterations = 50
                                (Bullet library default 10 = SolverPreset "bullet10")                              */
  float collision_margin;    /* convex-hull margin (pybullet URDF default 0.001)  */
  float erp;                 /* Baumgarte factor of a penetrating contact (Bullet m_erp = 0.2; applies down to the
                                split-impulse threshold of -0.04 m, which no resting contact reaches)              */
  float friction_rock;       /* lateral_friction of a rock (template.urdf: 0.6)   */
  float friction_ground;     /* Bullet default body friction 0.5                  */
  float linear_damping;      /* pybullet default 0.04                             */
  float angular_damping;     /* pybullet default 0.04                             */
  float warmstart;           /* m_warmstartingFactor: pybullet server 0.1 (Bullet library default 0.85)             */
  float linear_slop;         /* m_linearSlop: pybullet server 1e-5 (library default 0): penetration = distance + slop */
  float residual_threshold;  /* m_leastSquaresResidualThreshold: pybullet server 1e-7 (library default 0 = never): the
                                sweeps of a sub-step end once max over rows of (delta impulse x effective-mass
                                denominator)^2 of a sweep is <= this                                               */
  int32_t place_at_com;      /* 1: resetBasePositionAndOrientation moves the COM frame (reference quirk) */
  /* --- TestStackEnv (Stack-v2), env.py:443-470 --- */
  int32_t orientation_freedom; /* k: the pending rock is observed in 2^k yaw orientations (observer.py:127-140) and the
                                  action chooses one: action = orientation * A + pixel; 0 = Stack-v0 (one orientation) */
  int32_t ordering_freedom;    /* 1: all rocks of the episode are shown from the start and the action also chooses which
                                  one to place (env.py:443-470, :496-506; simulator.py:343-378): the observation holds
                                  episode_length * 2^k object maps, the maps of the n rocks still unplaced first (rock-major,
                                  orientation-minor, observer.py:310-327), empty maps after them;
                                  action = (rock * 2^k + orientation) * A + pixel with rock < n; 0 = Stack-v0 */
} srl_config;

#ifdef __cplusplus
}
#endif
#endif /* SRL_TYPES_H_ */
