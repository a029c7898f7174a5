#!/usr/bin/env python3
"""Golden vectors for the reward / goal / observation-packing / episode-machine rows (R1, R2, E1, E4)
from the reference's OWN code: `stackrl/envs/stack/rewarder.py`, `stackrl/envs/stack/env.py` and
`stackrl/envs/stack/observer.py` are loaded by file path and run as they are.

Runs only in the build container (needs /root/reference).  The three files are numpy code below their import lines; the
packages they import but which are not installed get inert placeholders, as `make_baselines_golden.py` does for
`baselines.py`:

  gym                      `Env` (empty base class), `spaces.Box / Tuple / Discrete / MultiDiscrete` (records of their
                           arguments; `Discrete.contains`), `envs.registry` (unused here), `utils.seeding.np_random`
  stackrl.envs.data        `generated()` -> a list of made-up descriptor names (the env only samples and passes them on)
  stackrl.envs.stack.simulator
                           `Simulator` = the scripted stand-in below.  pybullet is what is absent, so everything pybullet
                           would compute is an INPUT of the fixture: the depth buffers both cameras return, where each rock
                           ends up, and how far it is from where it was placed.  The stand-in replays seeded synthetic
                           values and records what the reference's code asks of it (the pose it is told to place at).

Every random draw of `Rewarder._reset_goal` comes from an explicit list (`ScriptedDraws`), recorded next to the goal
rectangle the reference builds from it.

The file written (`rewarder_golden.npz`) holds data only: per case the configuration, per call the inputs (action, depth
buffers, rock positions, distances, goal draws) and what the reference returned (observation bytes, reward(s), done,
the pose `Observer.pose` asked for, the goal rectangle).  tests/test_rewarder_golden.py replays the inputs through the
oracle's restatement of the same functions.
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True   # the reference tree is read-only: no __pycache__ beside its files
import types

import numpy as np

REF = '/root/reference/stackrl/envs/stack/'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rewarder_golden.npz')


# ------------------------------------------------------------------------------------------------ placeholders
class ScriptedDraws(object):
  """Stands in for the `RandomState` of `Rewarder` (rewarder.py:103): every draw `_reset_goal` makes comes from an explicit
  list of (swap bit, Beta value as a 24-bit fraction, 32-bit word of the row offset, 32-bit word of the column offset)."""

  def __init__(self, gen):
    self.gen = gen                  # yields the draw lists, one per goal
    self.cur = None
    self.log = []                   # one record per goal: bbit, x24, ru, rv, the arguments beta() was called with

  def randint(self, a, b=None):
    if b is None:
      assert a == 2                 # rewarder.py:227: the swap of the Beta parameters
      self.cur = dict(zip(('bbit', 'x24', 'ru', 'rv'), next(self.gen)))
      self.cur['stage'] = 0
      return self.cur['bbit']
    r = self.cur['ru'] if self.cur['stage'] == 0 else self.cur['rv']
    self.cur['stage'] += 1
    if self.cur['stage'] == 2:
      self.log.append((self.cur['bbit'], self.cur['x24'], self.cur['ru'], self.cur['rv'], self.cur['beta_a'], self.cur['beta_b']))
    return a + ((r * (b - a)) >> 32)      # a uniform 32-bit word reduced to [a, b)

  def beta(self, a, b):
    self.cur['beta_a'], self.cur['beta_b'] = a, b
    return self.cur['x24'] / float(1 << 24)

  def seed(self, s):
    pass


class StubSimulator(object):
  """What `StackEnv`, `Observer` and `Rewarder` use of `Simulator` (simulator.py:63-128, :156-258), with pybullet's part
  replaced by a script: depth buffers are smooth seeded bumps where the rocks are, a placed rock ends up a seeded few
  millimetres from where `Observer.pose` put it, earlier rocks drift a little at every step."""
  script_rng = None       # set by the generator before the env is built
  res = None              # (overhead, object) resolutions

  def __init__(self, use_gui=False, time_step=0.01, gravity=9.8, spawn_position=(0, 0, 2), spawn_orientation=(0, 0, 0, 1),
               num_steps=None, velocity_threshold=0.01):
    self.ctor = dict(time_step=time_step, gravity=gravity, spawn_position=tuple(spawn_position),
                     spawn_orientation=tuple(spawn_orientation), num_steps=num_steps, velocity_threshold=velocity_threshold)
    self.new_pose = (tuple(spawn_position), tuple(spawn_orientation))
    self._has_new = False
    self.pending = None
    self.final = []       # [(position, orientation)]
    self.dist = []        # [(translation, rotation)]
    self.asked = []       # positions `step` was called with
    self.loaded = []      # descriptor names in the order they were loaded
    self.depths = []      # depth buffers handed out since the last take()

  # -- the pybullet calls Observer makes (observer.py:84-119, :143-164, :252-277)
  def computeViewMatrix(self, **kw):
    return ('view', kw)

  def computeProjectionMatrix(self, **kw):
    return ('proj', kw)

  def getQuaternionFromEuler(self, e):
    return (0., 0., 0., 1.)

  def multiplyTransforms(self, pa, qa, pb, qb):
    return tuple(np.add(pa, pb)), qa

  def invertTransform(self, p, q):
    return tuple(-np.asarray(p)), q

  def getCameraImage(self, width, height, viewMatrix, projectionMatrix):
    rng = self.script_rng
    H, h = self.res
    if height == H:                       # overhead camera: depth 1 = nothing seen (elevation 0)
      d = np.ones((H, H), np.float32)
      ii, jj = np.mgrid[0:H, 0:H]
      for (p, _) in self.final:
        ci, cj = p[0] / 0.5 * H, p[1] / 0.5 * H
        r2 = ((ii - ci) / (H * 0.09)) ** 2 + ((jj - cj) / (H * 0.07)) ** 2
        d = np.minimum(d, (0.35 + 0.5 * np.minimum(r2, 1.3)).astype(np.float32) + np.float32(0.2) * (r2 > 1))
      d = np.clip(d + (rng.uniform(-0.004, 0.004, size=d.shape).astype(np.float32)) * (d < 1), 0, 1).astype(np.float32)
    else:                                 # object camera (from below): a bump when a rock is waiting, else nothing
      d = np.ones((h, h), np.float32)
      if self.pending is not None:
        ii, jj = np.mgrid[0:h, 0:h]
        a, b = rng.uniform(0.25, 0.45) * h, rng.uniform(0.2, 0.4) * h
        r2 = ((ii - h / 2 + 0.5) / a) ** 2 + ((jj - h / 2 + 0.5) / b) ** 2
        d = np.where(r2 <= 1, 0.3 + 0.4 * r2, 1.0).astype(np.float32)
    self.depths.append(d)
    return width, height, None, d, None

  # -- Simulator's own interface
  @property
  def has_new_object(self):               # simulator.py:70-77
    if self._has_new:
      self._has_new = False
      return True
    return False

  def _load(self, urdf):                  # simulator.py:297-308
    self.pending = urdf
    if urdf is not None:
      self.loaded.append(str(urdf))
      self._has_new = True

  def reset(self, urdf):                  # simulator.py:156-188
    self.final, self.dist = [], []
    self._load(urdf)

  def __call__(self, *args, **kwargs):
    return self.step(*args, **kwargs)

  def step(self, position, orientation=(0, 0, 0, 1), urdf=None, smooth_placing=False):
    rng = self.script_rng
    f32 = lambda x: float(np.float32(x))    # values a float32 oracle can be given exactly
    self.asked.append(tuple(float(c) for c in position))
    self.smooth = smooth_placing
    # earlier rocks drift, the new one settles near where it was asked to be
    for k in range(len(self.final)):
      p, q = self.final[k]
      self.final[k] = (tuple(f32(c + rng.normal(0, 0.002)) for c in p), q)
      dp, do = self.dist[k]
      self.dist[k] = (f32(dp + abs(rng.normal(0, 0.004))), f32(do + abs(rng.normal(0, 0.05))))
    p = (f32(position[0] + rng.normal(0, 0.01)), f32(position[1] + rng.normal(0, 0.01)), f32(max(position[2] - abs(rng.normal(0, 0.01)), 0.01)))
    self.final.append((p, tuple(orientation)))
    big = rng.uniform() < 0.25            # now and then a rock rolls far: discounts reach their clamp at 0
    self.dist.append((f32(abs(rng.normal(0, 0.15 if big else 0.01))), f32(abs(rng.normal(0, 3.0 if big else 0.1)))))
    self._load(urdf)

  @property
  def positions(self):                    # simulator.py:85-88
    return [p for p, _ in self.final]

  @property
  def distances_from_place(self):         # simulator.py:113-128
    return list(self.dist)

  def disconnect(self):
    pass

  def isConnected(self):
    return True


def load_reference(draws):
  """The reference's observer.py, rewarder.py and env.py as modules, over the placeholders."""
  gym = types.ModuleType('gym')

  class Env(object):
    pass
  gym.Env = Env
  spaces = types.ModuleType('gym.spaces')

  class _Space(object):
    def __init__(self, *a, **k):
      self.args, self.kwargs = a, k
      self.shape = k.get('shape')

  class Discrete(_Space):
    def __init__(self, n):
      _Space.__init__(self, n)
      self.n = int(n)

    def contains(self, x):
      return isinstance(x, (int, np.integer)) and 0 <= x < self.n
  spaces.Box = spaces.MultiDiscrete = _Space
  spaces.Discrete = Discrete

  class Tuple_(_Space):
    def __init__(self, spaces_):
      _Space.__init__(self, spaces_)
      self.spaces = tuple(spaces_)

    def __getitem__(self, i):
      return self.spaces[i]
  spaces.Tuple = Tuple_
  gym.spaces = spaces
  genvs = types.ModuleType('gym.envs')
  genvs.registry = object()
  gym.envs = genvs
  gutils = types.ModuleType('gym.utils')
  seeding = types.ModuleType('gym.utils.seeding')
  calls = {'n': 0}

  def np_random(seed=None):
    calls['n'] += 1
    # env.py:108 asks first (mesh choice, rewarder seed), rewarder.py:103 second (the goal draws)
    return (np.random.RandomState(seed if seed is None or seed < 2 ** 32 else seed % 2 ** 32) if calls['n'] % 2 == 1 else draws), seed
  seeding.np_random = np_random
  seeding.create_seed = lambda s=None: s
  gutils.seeding = seeding
  gym.utils = gutils

  stackrl = types.ModuleType('stackrl')
  senvs = types.ModuleType('stackrl.envs')
  data = types.ModuleType('stackrl.envs.data')
  data.generated = lambda **kw: ['rock_{:03d}'.format(i) for i in range(40)]
  senvs.data = data
  sstack = types.ModuleType('stackrl.envs.stack')
  ssim = types.ModuleType('stackrl.envs.stack.simulator')
  ssim.Simulator = StubSimulator
  ssim.TestSimulator = StubSimulator
  mods = {'gym': gym, 'gym.spaces': spaces, 'gym.envs': genvs, 'gym.utils': gutils, 'gym.utils.seeding': seeding,
          'stackrl': stackrl, 'stackrl.envs': senvs, 'stackrl.envs.data': data, 'stackrl.envs.stack': sstack,
          'stackrl.envs.stack.simulator': ssim}
  saved = {k: sys.modules.get(k) for k in list(mods) + ['stackrl.envs.stack.observer', 'stackrl.envs.stack.rewarder']}
  sys.modules.update(mods)
  out = {}
  try:
    for name in ('observer', 'rewarder', 'env'):
      spec = importlib.util.spec_from_file_location('stackrl.envs.stack.' + name, REF + name + '.py')
      mod = importlib.util.module_from_spec(spec)
      sys.modules['stackrl.envs.stack.' + name] = mod
      spec.loader.exec_module(mod)
      out[name] = mod
  finally:
    for k, v in saved.items():
      if v is None:
        sys.modules.pop(k, None)
      else:
        sys.modules[k] = v
    sys.modules.pop('stackrl.envs.stack.env', None)
  return out


CASES = [
  # tag, StackEnv kwargs (the Stack-v0 registry passes dtype='uint8', goal_size_ratio=.25, reward_params=2), calls
  ('a', dict(episode_length=4, resolution_factor=5), 8),                                             # Stack-v0 defaults: IoU, scale 1
  ('b', dict(episode_length=3, resolution_factor=5, rewarder='all', reward_scale=None, reward_params=(1, 3)), 6),
  ('c', dict(episode_length=3, resolution_factor=4, rewarder='eval', reward_scale=2.5, reward_params=(2, 1)), 6),
  ('d', dict(episode_length=4, resolution_factor=4, rewarder='dor', reward_scale=None, sim_time_step=0.0125), 7),   # config.gin overrides
  ('e', dict(episode_length=3, resolution_factor=4, rewarder='diou', reward_scale=1.0), 9),
  ('f', dict(episode_length=3, resolution_factor=4, rewarder='or', reward_scale=0.5, goal_size_ratio=0.1), 9),
  ('g', dict(episode_length=2, resolution_factor=4, rewarder='DOR', goal_size_ratio=0.5, reward_params=None), 6),
]


def goal_draws(rng):
  # end points of every range first, then seeded draws
  fixed = [(0, 0, 0, 0), (1, (1 << 24) - 1, 0xffffffff, 0xffffffff), (0, (1 << 24) - 1, 0, 0xffffffff), (1, 0, 0xffffffff, 0)]
  for f in fixed:
    yield f
  while True:
    yield (int(rng.randint(2)), int(rng.randint(1 << 24)), int(rng.randint(0, 2 ** 32, dtype=np.uint64)), int(rng.randint(0, 2 ** 32, dtype=np.uint64)))


def main():
  rng = np.random.RandomState(11)
  draws = ScriptedDraws(goal_draws(rng))
  ref = load_reference(draws)
  StackEnv, Rewarder = ref['env'].StackEnv, ref['rewarder'].Rewarder
  out = {'cases': np.array([c[0] for c in CASES])}
  for tag, kw, ncalls in CASES:
    StubSimulator.script_rng = rng
    h = 2 ** kw['resolution_factor']
    StubSimulator.res = (4 * h, h)
    full = dict(dtype='uint8', goal_size_ratio=.25, reward_params=2, seed=5)     # envs/stack/__init__.py:4-8
    full.update(kw)
    env = StackEnv(**full)
    sim = env._sim
    A = env.action_space.n
    rec = {k: [] for k in ('action', 'd_over', 'd_obj', 'npos', 'pos', 'dist', 'obs_map', 'obs_obj', 'reward', 'done',
                           'asked', 'was_reset', 'goal')}
    ngoals0 = len(draws.log)
    for call in range(ncalls):
      a = int(rng.randint(A))
      (gu0, gv0), (gu1, gv1) = env._rew._goal_lims
      if gu1 > gu0 and rng.uniform() < 0.75:        # mostly aim at the goal, so that the discounted metrics are exercised
        AW = 3 * h + 1
        a = int(np.clip(rng.randint(gu0, gu1) - h // 2, 0, AW - 1)) * AW + int(np.clip(rng.randint(gv0, gv1) - h // 2, 0, AW - 1))
      if tag == 'a' and call == 1:
        a = A - 1                                   # the last action: pixel (AW - 1, AW - 1)
      if tag == 'a' and call == 2:
        a = 0
      sim.depths = []
      n_asked = len(sim.asked)
      was_reset = bool(env._done)                   # env.py:235-236: this call is the (auto-)reset
      res = env.step(a)
      if was_reset:
        obs, reward, done, info = res
        assert reward == 0. and done is False and info == {}
      else:
        obs, reward, done, info = res
      keys = [k for k in ('IoU', 'OR', 'DIoU', 'DOR', 'AD') if isinstance(info, dict) and k in info]
      rv = [float(reward)] if reward is not None else [float(info[k]) for k in keys]
      width = {'all': 4, 'eval': 2}.get(str(full.get('rewarder')).lower(), 1)
      rv = rv + [0.0] * (width - len(rv))           # the reset call returns the scalar 0.0 whatever the metric
      assert len(sim.depths) == 2 and sim.depths[0].shape[0] == 4 * h
      rec['action'].append(a)
      rec['d_over'].append(sim.depths[0]); rec['d_obj'].append(sim.depths[1])
      pos = np.zeros((8, 3), np.float32); dist = np.zeros((8, 2), np.float32)
      for k, p in enumerate(sim.positions):
        pos[k] = p; dist[k] = sim.distances_from_place[k]
        assert tuple(float(c) for c in pos[k]) == tuple(p) and tuple(float(c) for c in dist[k]) == tuple(sim.distances_from_place[k])
      rec['npos'].append(len(sim.positions)); rec['pos'].append(pos); rec['dist'].append(dist)
      rec['obs_map'].append(np.asarray(obs[0])); rec['obs_obj'].append(np.asarray(obs[1]))
      assert rec['obs_map'][-1].dtype == np.uint8 and rec['obs_map'][-1].shape == (4 * h, 4 * h, 2)
      rec['reward'].append(rv); rec['done'].append(bool(done)); rec['was_reset'].append(was_reset)
      rec['asked'].append(sim.asked[-1] if len(sim.asked) > n_asked else (np.nan,) * 3)
      (u0, v0), (u1, v1) = env._rew._goal_lims
      rec['goal'].append((u0, v0, u1 - u0, v1 - v0))
      assert np.array_equal(env._rew.goal != 0, env._rew.goal_bin) and float(env._rew.goal.max()) == float(np.float32(env._obs.max_z))
    out[tag + '_kwargs'] = np.array(repr(sorted(full.items())))
    out[tag + '_n_actions'] = np.array(A)
    out[tag + '_scale'] = np.array(env._rew.scale)
    out[tag + '_goal_draws'] = np.array(draws.log[ngoals0:], dtype=np.int64)      # one row per episode, in order
    out[tag + '_loaded'] = np.array(sim.loaded)
    for k, v in rec.items():
      dt = {'action': np.int64, 'npos': np.int32, 'reward': np.float64, 'done': np.bool_, 'was_reset': np.bool_, 'asked': np.float64,
            'goal': np.int32}.get(k)
      out[tag + '_' + k] = np.array(v, dtype=dt) if dt is not None else np.stack(v)
    env.close()

  # ---- the goal rectangle alone, over many draw lists and geometries (rewarder.py:211-259)
  goals = []
  obs_mod = ref['observer']
  for (H, h, ratio) in ((128, 32, 0.25), (128, 32, 0.5), (128, 32, 0.1), (64, 16, 0.25), (64, 16, 1.0), (96, 32, 0.3)):
    sim = StubSimulator(spawn_position=(0, 0, 0.5))
    StubSimulator.res = (H, h)
    obs = obs_mod.Observer(sim, overhead_resolution=H, object_resolution=h, pixel_size=0.125 / h, max_z=0.375)
    d2 = ScriptedDraws(goal_draws(np.random.RandomState(H + int(100 * ratio))))
    saved = sys.modules.get('gym.utils.seeding')
    r = Rewarder.__new__(Rewarder)
    # the constructor as it is, with the draws' source handed in through the seeding placeholder
    ref['rewarder'].seeding.np_random = lambda seed=None, _d=d2: (_d, seed)
    r.__init__(sim, obs, goal_size_ratio=ratio, n_objects=8)
    for _ in range(40):
      r.reset()
      (u0, v0), (u1, v1) = r._goal_lims
      b = d2.log[-1]
      goals.append((H, h, int(round(ratio * 1000)), b[1], b[2], b[3], b[0], b[4], b[5], u0, v0, u1 - u0, v1 - v0,
                    int(round(float(r._goal_volume) * 1e6))))
  out['goals'] = np.array(goals, dtype=np.int64)
  out['goals_columns'] = np.array('H h ratio_x1000 x24 ru rv swap_bit beta_a beta_b u v gh gw volume_x1e6')
  np.savez_compressed(OUT, **out)
  print('wrote', OUT, os.path.getsize(OUT), 'bytes')
  for k in sorted(out):
    if k.startswith('a_') or k in ('goals',):
      print(' ', k, getattr(out[k], 'shape', None), getattr(out[k], 'dtype', None))


if __name__ == '__main__':
  main()
