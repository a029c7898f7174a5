"""Flat configuration of the Stack-v0 hot path and its ctypes mirror.

Carries the resolved parameters of `stackrl/envs/stack/env.py:28-51` (class defaults) over the
gym-registry kwargs of `stackrl/envs/stack/__init__.py:4-8` (Stack-v0: `urdfs='[5-9]?'`,
`reward_params=2`, `dtype='uint8'`).  gin itself is out of scope;
the values of `config.gin` that BASELINE config 1 needs are exposed by `StackConfig.config_gin()`.
"""
import ctypes
import dataclasses

ACTION_HOLD = -2      # SRL_ACTION_HOLD (include/srl_types.h)
MAX_BODIES = 32
MAX_VERTS = 128
MAX_TRIS = 252

# Named solver definitions: 'pybullet' = the defaults above (what the reference runs: it only calls setTimeStep,
# simulator.py:143); 'bullet10' = the Bullet library defaults (btContactSolverInfo) that round 1 was measured on.
SOLVER_PRESETS = {
  'pybullet': dict(solver_iterations=50, warmstart=0.1, linear_slop=1e-5, residual_threshold=1e-7),
  'bullet10': dict(solver_iterations=10, warmstart=0.85, linear_slop=0.0, residual_threshold=0.0),
}

METRICS = {'iou': 0, 'or': 1, 'diou': 2, 'dor': 3, 'all': 4, 'eval': 5}  # rewarder.py:7-14
REWARD_KEYS = {4: ('IoU', 'OR', 'DIoU', 'DOR'), 5: ('IoU', 'AD')}      # the dict keys of rewarder.py:147-158

# return codes, include/srl_types.h
OK, EINVAL, EINVAL_ACTION, ESIM_DIVERGED, EHIP, ENOMESH = range(6)
ST_DIVERGED, ST_PAIR_OVERFLOW, ST_BAD_ACTION = 1, 2, 4


class CConfig(ctypes.Structure):
  """Mirror of `srl_config` (include/srl_types.h)."""
  _fields_ = [
    ('n_envs', ctypes.c_int32),
    ('env_index_offset', ctypes.c_int32),
    ('episode_length', ctypes.c_int32),
    ('overhead_res', ctypes.c_int32),
    ('object_res', ctypes.c_int32),
    ('object_max_dimension', ctypes.c_float),
    ('max_z', ctypes.c_float),
    ('sim_time_step', ctypes.c_float),
    ('gravity', ctypes.c_float),
    ('velocity_threshold', ctypes.c_float),
    ('smooth_placing', ctypes.c_int32),
    ('max_substeps', ctypes.c_int32),
    ('metric', ctypes.c_int32),
    ('goal_size_ratio', ctypes.c_float),
    ('reward_scale', ctypes.c_float),
    ('reward_pexp', ctypes.c_int32),
    ('reward_oexp', ctypes.c_int32),
    ('solver_iterations', ctypes.c_int32),
    ('collision_margin', ctypes.c_float),
    ('erp', ctypes.c_float),
    ('friction_rock', ctypes.c_float),
    ('friction_ground', ctypes.c_float),
    ('linear_damping', ctypes.c_float),
    ('angular_damping', ctypes.c_float),
    ('warmstart', ctypes.c_float),
    ('linear_slop', ctypes.c_float),
    ('residual_threshold', ctypes.c_float),
    ('place_at_com', ctypes.c_int32),
    ('orientation_freedom', ctypes.c_int32),
    ('ordering_freedom', ctypes.c_int32),
  ]


@dataclasses.dataclass
class StackConfig:
  """Stack-v0 parameters (env.py:28-51) + the build-owned solver definition."""
  n_envs: int = 1
  env_index_offset: int = 0
  episode_length: int = 30            # DEFAULT_EPISODE_LENGTH, env.py:20
  object_max_dimension: float = 0.125
  sim_time_step: float = 1 / 100.
  gravity: float = 9.8
  num_sim_steps: object = None        # only None works in the reference (simulator.py:325-326 quirk)
  velocity_threshold: float = 0.01
  smooth_placing: bool = True
  observable_size_ratio: int = 4
  resolution_factor: int = 5
  max_z: float = 0.375
  rewarder: object = None             # None -> IoU, rewarder.py:113-114
  goal_size_ratio: float = .25
  reward_scale: object = 1.           # None -> n_objects, rewarder.py:97
  reward_params: object = 2           # Stack-v0 registry kwarg
  flat_action: bool = True
  dtype: str = 'uint8'                # Stack-v0 registry kwarg
  max_substeps: int = 0               # 0 -> int(300/time_step), simulator.py:46
  # solver definition: what pybullet's physics server runs with when the client only sets the
  # time step (simulator.py:143).  SOLVER_PRESETS['bullet10'] holds the Bullet *library* defaults of round 1.
  solver_iterations: int = 50         # pybullet numSolverIterations default
  collision_margin: float = 0.001
  erp: float = 0.2
  friction_rock: float = 0.6          # template.urdf lateral_friction, generator.py:250
  friction_ground: float = 0.5
  linear_damping: float = 0.04
  angular_damping: float = 0.04
  warmstart: float = 0.1              # pybullet server m_warmstartingFactor
  linear_slop: float = 1e-5           # pybullet server m_linearSlop
  residual_threshold: float = 1e-7    # pybullet server m_leastSquaresResidualThreshold (0 = no early exit)
  place_at_com: bool = True
  orientation_freedom: int = 0        # TestStackEnv (Stack-v2, env.py:443-470): 2**k yaw orientations; 0 = Stack-v0
  ordering_freedom: bool = False      # TestStackEnv: all rocks on show from the start, the action picks the next one

  def __post_init__(self):
    if not 0 <= int(self.orientation_freedom) <= 4:
      raise ValueError('orientation_freedom must be in 0..4 (at most 16 orientations)')
    if self.dtype != 'uint8':
      # env.py:169-170 raises ValueError for unknown dtypes; the build implements the Stack-v0 one.
      raise ValueError('Invalid value {} for argument dtype.'.format(self.dtype))
    if self.num_sim_steps:
      raise ValueError('num_sim_steps is not supported (the reference branch is broken, simulator.py:325-326).')
    if not self.flat_action:
      raise ValueError('Only flat_action=True (Stack-v0) is implemented.')
    if not (1 <= self.episode_length <= MAX_BODIES):
      raise ValueError('episode_length must be in [1, {}].'.format(MAX_BODIES))
    self.exponents()   # rewarder.py:129-142 argument checks
    self.metric_id     # rewarder.py:116-121

  @property
  def object_res(self):
    return 2 ** self.resolution_factor                      # env.py:128

  @property
  def overhead_res(self):
    return self.object_res * self.observable_size_ratio     # env.py:129-130

  @property
  def pixel_size(self):
    return self.object_max_dimension / self.object_res      # env.py:136

  @property
  def n_actions(self):
    return (self.overhead_res - self.object_res + 1) ** 2   # env.py:207-211

  @property
  def n_orientations(self):
    return 2 ** int(self.orientation_freedom)               # observer.py:127

  @property
  def n_object_maps(self):
    """Object maps per observation: the orientations of the pending rock, or with ordering freedom those of every rock
    of the episode (env.py:472-480; the maps of placed rocks are empty and sit at the end)."""
    return self.n_orientations * (self.episode_length if self.ordering_freedom else 1)

  @property
  def metric_id(self):
    m = self.rewarder
    if m is None:
      return 0
    if isinstance(m, str):
      m = {'position': 'dor', 'occupation': 'or'}.get(m, m)  # env.py:148-155
      if m.lower() not in METRICS:
        raise ValueError('Invalid value {} for argument metric.'.format(m))
      return METRICS[m.lower()]
    if m not in (0, 1, 2, 3, 4, 5):
      raise ValueError('Invalid value {} for argument metric.'.format(m))
    return int(m)

  @property
  def reward_keys(self):
    """None for a scalar reward; for 'all' / 'eval' the keys of the dict the reference returns (rewarder.py:147-158) —
    here the columns of the reward tensor [B, K]."""
    return REWARD_KEYS.get(self.metric_id)

  def exponents(self):
    p = self.reward_params                                   # rewarder.py:129-142
    if p is None:
      return -1, -1
    if isinstance(p, (int, float)):
      p = (p, p)
    elif len(p) == 1:
      p = (p[0], p[0])
    pe, oe = p[0], p[1]
    if pe < 0 or oe < 0:
      raise ValueError('Invalid value {} for argument params. Must be non negative.'.format(p))
    if int(pe) != pe or int(oe) != oe:
      raise ValueError('reward_params must be integers in this build.')
    return int(pe), int(oe)

  def to_c(self):
    pe, oe = self.exponents()
    return CConfig(
      n_envs=self.n_envs, env_index_offset=self.env_index_offset,
      episode_length=self.episode_length, overhead_res=self.overhead_res,
      object_res=self.object_res, object_max_dimension=self.object_max_dimension,
      max_z=self.max_z, sim_time_step=self.sim_time_step, gravity=self.gravity,
      velocity_threshold=self.velocity_threshold, smooth_placing=int(bool(self.smooth_placing)),
      # simulator.py:46 `int(MAX_STEP_TIME / time_step)` in Python's doubles, like the reference (the C struct carries the time
      # step as a float32: 300 / float32(0.0125) = 23999.9996, one step short of the reference's 24000)
      max_substeps=int(self.max_substeps) or int(300 / self.sim_time_step), metric=self.metric_id,
      goal_size_ratio=self.goal_size_ratio,
      reward_scale=-1. if self.reward_scale is None else float(self.reward_scale),
      reward_pexp=pe, reward_oexp=oe, solver_iterations=self.solver_iterations,
      collision_margin=self.collision_margin, erp=self.erp, friction_rock=self.friction_rock,
      friction_ground=self.friction_ground, linear_damping=self.linear_damping,
      angular_damping=self.angular_damping, warmstart=self.warmstart,
      linear_slop=self.linear_slop, residual_threshold=self.residual_threshold,
      place_at_com=int(bool(self.place_at_com)),
      orientation_freedom=int(self.orientation_freedom),
      ordering_freedom=int(bool(self.ordering_freedom)),
    )

  @classmethod
  def config_gin(cls, **kw):
    """Env overrides of the root `config.gin:4-17` (BASELINE config 1)."""
    base = dict(sim_time_step=0.0125, rewarder='dor', reward_scale=None)
    base.update(kw)
    return cls(**base)
