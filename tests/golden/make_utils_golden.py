#!/usr/bin/env python3
"""Golden vectors for the batching boundary from the reference's OWN code:
`stackrl/envs/utils.py` — `Env` and `ParallelEnv` with its worker processes and pipes — is loaded by file path and run as it is.

Runs only in the build container (needs /root/reference).  utils.py imports gin, gym and tensorflow, none of which is installed;
it uses very little of them, so each gets an inert placeholder:

  gin          `configurable(**kw)` -> the identity decorator
  gym          `Env`, `Space` (base classes), `make(id, **kwargs)` -> the scripted env below; its `Tuple` / `Box` / `Discrete` spaces
               carry `shape`, `dtype`, `sample()`, `seed()`
  tensorflow   `constant`, `zeros`, `unstack`, `TensorSpec`, `float32` / `bool`, `nest.{is_nested, map_structure, flatten,
               pack_sequence_as}` over numpy arrays (a `Tensor` is an ndarray with `.numpy()`)

The scripted env stands in for `StackEnv`: its observations, rewards and done flags are a deterministic function of (the seed
it was MADE or re-seeded with, the episode, the step, the action it was handed), so the batched tensors the reference returns
show which worker got which seed and which action, in which order they were stacked, with which dtypes — which is what B1 is.

The file written (`utils_golden.npz`) holds data only: per session the constructor arguments and per call what the reference
returned.  tests/test_utils_golden.py compares it with this repo's statement of the same semantics.
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True   # the reference tree is read-only: no __pycache__ beside its files
import types

import numpy as np

REF = '/root/reference/stackrl/envs/utils.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'utils_golden.npz')
H, h, A, L = 8, 4, 25, 3        # toy sizes: map H x H x 2, object map h x h x 1, A actions, L steps per episode


def mix(*xs):
  v = 0x9E3779B9
  for x in xs:
    v = ((v ^ (int(x) & 0xffffffff)) * 0x85EBCA6B + 0xC2B2AE35) & 0xffffffff
    v ^= v >> 15
  return v


def scripted_step(seed, episode, t, action):
  """(obs_map u8 [H,H,2], obs_obj u8 [h,h,1], reward float, done bool) of the scripted env."""
  k = mix(seed, episode, t, action)
  om = ((np.arange(H * H * 2, dtype=np.uint32) * 2654435761 + k) >> 13).astype(np.uint8).reshape(H, H, 2)
  oo = ((np.arange(h * h, dtype=np.uint32) * 40503 + k) >> 7).astype(np.uint8).reshape(h, h, 1)
  return om, oo, float((k % 1000) / 1000.0 - 0.25), bool(t >= L)


# ------------------------------------------------------------------------------------------------ placeholders
def make_gin():
  gin = types.ModuleType('gin')
  gin.configurable = lambda *a, **kw: (lambda f: f)
  return gin


def make_gym():
  gym = types.ModuleType('gym')

  class Space(object):
    def seed(self, s=None):
      self._rng = np.random.RandomState(None if s is None else int(s) % (2 ** 32))
      return [s]

  class Box(Space):
    def __init__(self, shape, dtype):
      self.shape, self.dtype = tuple(shape), np.dtype(dtype)
      self.seed(0)

    def sample(self):
      return self._rng.randint(0, 256, size=self.shape).astype(self.dtype)

  class Tuple(Space):
    def __init__(self, spaces):
      self.spaces = tuple(spaces)

    def sample(self):
      return tuple(s.sample() for s in self.spaces)

  class Discrete(Space):
    def __init__(self, n):
      self.n, self.shape, self.dtype = n, (), np.dtype(np.int64)
      self.seed(0)

    def sample(self):
      return int(self._rng.randint(self.n))

  class Env(object):
    pass

  class ScriptedEnv(Env):
    def __init__(self, seed=None, **kwargs):
      self.kwargs = dict(kwargs)
      self.observation_space = Tuple((Box((H, H, 2), np.uint8), Box((h, h, 1), np.uint8)))
      self.action_space = Discrete(A)
      self.seed(seed)

    def seed(self, seed=None):
      self._seed = 0 if seed is None else int(seed)
      self._episode, self._t = 0, 0
      return [self._seed, mix(self._seed, 77)]          # (StackEnv.seed returns its seed and the rewarder's, env.py:341-346)

    def reset(self):
      self._episode += 1
      self._t = 0
      om, oo, _, _ = scripted_step(self._seed, self._episode, 0, -1)
      return om, oo

    def step(self, action):
      self._t += 1
      om, oo, r, d = scripted_step(self._seed, self._episode, self._t, int(action))
      return (om, oo), r, d, {'ignored': True}

    def render(self, mode='human'):
      return ('frame', self._seed, mode)

    def close(self):
      pass

  gym.Env, gym.Space, gym.make = Env, Space, (lambda env_id, **kwargs: ScriptedEnv(**kwargs))
  return gym


def make_tf():
  tf = types.ModuleType('tensorflow')

  class Tensor(np.ndarray):
    def numpy(self):
      return np.asarray(self)

  def constant(value, dtype=None):
    return np.asarray(value, dtype=dtype).view(Tensor)

  class TensorSpec(object):
    def __init__(self, shape, dtype):
      self.shape, self.dtype = tuple(int(v) for v in shape), np.dtype(dtype)

  nest = types.ModuleType('tensorflow.nest')
  nest.is_nested = lambda v: isinstance(v, (tuple, list, dict))

  def map_structure(fn, *structs):
    s0 = structs[0]
    if nest.is_nested(s0):
      return tuple(map_structure(fn, *[s[i] for s in structs]) for i in range(len(s0)))
    return fn(*structs)

  def flatten(s):
    return [x for y in s for x in flatten(y)] if nest.is_nested(s) else [s]

  def pack_sequence_as(spec, flat):
    it = iter(flat)
    return map_structure(lambda _: next(it), spec)

  nest.map_structure, nest.flatten, nest.pack_sequence_as = map_structure, flatten, pack_sequence_as
  tf.constant, tf.TensorSpec, tf.nest = constant, TensorSpec, nest
  tf.zeros = lambda shape, dtype=None: np.zeros(shape, dtype=dtype).view(Tensor)
  tf.unstack = lambda t: [np.asarray(x).view(Tensor) for x in t]
  tf.float32, tf.bool = np.float32, np.bool_
  return tf


def load_reference():
  sys.modules['gin'], sys.modules['gym'], sys.modules['tensorflow'] = make_gin(), make_gym(), make_tf()
  spec = importlib.util.spec_from_file_location('ref_utils', REF)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def main():
  ref = load_reference()
  tf = sys.modules['tensorflow']
  out = {'H': H, 'h': h, 'A': A, 'L': L}
  rng = np.random.RandomState(5)
  # ---- ParallelEnv: three workers, non-blocking by default (block=None -> False), then blocking
  for tag, (n, seed, block) in {'p0': (3, 11, None), 'p1': (2, 2 ** 32 - 1, True)}.items():
    env = ref.ParallelEnv('Scripted-v0', n_parallel=n, block=block, seed=seed)
    out[tag + '_n'], out[tag + '_seed'], out[tag + '_block'] = n, seed, -1 if block is None else int(block)
    out[tag + '_batch_size'], out[tag + '_multiprocessing'] = env.batch_size, int(env.multiprocessing)
    out[tag + '_obs_spec_shapes'] = np.asarray([str(tuple(s.shape)) for s in env.observation_spec])
    out[tag + '_obs_spec_dtypes'] = np.asarray([str(s.dtype) for s in env.observation_spec])
    out[tag + '_action_spec'] = np.asarray([str(env.action_spec.dtype), str(env.action_spec.shape)])
    r = env.reset()
    out[tag + '_reset_is_callable'] = int(callable(r))
    r = r() if callable(r) else r
    calls = [r]
    actions = []
    for t in range(2 * L + 1):                       # through `done` and on (the wrapper itself never resets)
      a = rng.randint(0, A, size=n).astype(np.int64)
      actions.append(a)
      s = env.step(tf.constant(a, dtype=np.int64))
      if t == 0:
        out[tag + '_step_is_callable'] = int(callable(s))
      calls.append(s() if callable(s) else s)
      if t == L:                                     # the caller resets (training.py:401); workers start a new episode
        rr = env.reset()
        calls.append(rr() if callable(rr) else rr)
        actions.append(np.full(n, -1, np.int64))
    out[tag + '_actions'] = np.stack(actions)
    for k, ((om, oo), rew, done) in enumerate(calls):
      out['{}_c{}_om'.format(tag, k)], out['{}_c{}_oo'.format(tag, k)] = np.asarray(om), np.asarray(oo)
      out['{}_c{}_r'.format(tag, k)], out['{}_c{}_d'.format(tag, k)] = np.asarray(rew), np.asarray(done)
    out[tag + '_n_calls'] = len(calls)
    out[tag + '_reseed'] = np.asarray(env.seed(1000), dtype=np.int64)          # list of the workers' seed() returns
    (om, oo), _, _ = env.reset(block=True)
    out[tag + '_after_reseed_om'] = np.asarray(om)
    smp = [np.asarray(env.sample()) for _ in range(3)]
    out[tag + '_samples'] = np.stack(smp)
    out[tag + '_sample_dtype'] = str(smp[0].dtype)
    out[tag + '_render'] = np.asarray([str(x) for x in env.render('rgb_array')])
    env.terminate()
  # ---- Env: the single-process wrapper (batch of one)
  env = ref.Env('Scripted-v0', seed=4)
  (om, oo), rew, done = env.reset()
  out['e_reset_om'], out['e_reset_r'], out['e_reset_d'] = np.asarray(om), np.asarray(rew), np.asarray(done)
  (om, oo), rew, done = env.step(tf.constant([7], dtype=np.int64))
  out['e_step_om'], out['e_step_oo'], out['e_step_r'], out['e_step_d'] = np.asarray(om), np.asarray(oo), np.asarray(rew), np.asarray(done)
  out['e_batch_size'], out['e_multiprocessing'] = env.batch_size, int(env.multiprocessing)
  out['e_sample_shape'] = np.asarray(np.asarray(env.sample()).shape)
  np.savez_compressed(OUT, **{k: np.asarray(v) for k, v in out.items()})
  print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
  main()
