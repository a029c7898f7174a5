"""The batching boundary B1 (`Env` utils.py:185-300, `ParallelEnv` utils.py:302-576) pinned by the reference's own
code: tests/golden/utils_golden.npz holds what the reference's `ParallelEnv` / `Env` returned for a scripted gym env whose outputs
are a function of (seed, episode, step, action) — tests/golden/make_utils_golden.py made it, in the build container.

CPU: this repo's statement of the semantics (the few lines of `Batch` below: which worker gets which seed and which action, the
stacking order, the dtypes, the zero reward / done of a reset, no reset by the wrapper, callable results when non-blocking) reproduces
every recorded tensor bit for bit.
GPU: `VecStackEnv` shows the same interface facts (`-m gpu`): there the envs are `StackEnv`s, so only what does not depend on the
env's content is compared with the record.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'utils_golden.npz')


@pytest.fixture(scope='module')
def gold():
  return np.load(GOLD)


# ------------------------------------------------------------------------------------------------ the scripted env (as the generator's)
def mix(*xs):
  v = 0x9E3779B9
  for x in xs:
    v = ((v ^ (int(x) & 0xffffffff)) * 0x85EBCA6B + 0xC2B2AE35) & 0xffffffff
    v ^= v >> 15
  return v


class Scripted(object):
  def __init__(self, g, seed):
    self.H, self.h, self.L = int(g['H']), int(g['h']), int(g['L'])
    self.seed(seed)

  def seed(self, seed):
    self._seed, self._episode, self._t = int(seed), 0, 0
    return [self._seed, mix(self._seed, 77)]

  def _out(self, action):
    H, h = self.H, self.h
    k = mix(self._seed, self._episode, self._t, action)
    om = ((np.arange(H * H * 2, dtype=np.uint32) * 2654435761 + k) >> 13).astype(np.uint8).reshape(H, H, 2)
    oo = ((np.arange(h * h, dtype=np.uint32) * 40503 + k) >> 7).astype(np.uint8).reshape(h, h, 1)
    return (om, oo), float((k % 1000) / 1000.0 - 0.25), bool(self._t >= self.L)

  def reset(self):
    self._episode += 1
    self._t = 0
    return self._out(-1)[0]

  def step(self, action):
    self._t += 1
    return self._out(int(action))


# ------------------------------------------------------------------------------------------------ B1 as this repo states it
class Batch(object):
  """`ParallelEnv` in a dozen lines: worker i is made with (seed + i) % 2**32 (utils.py:431-433) and receives action[i]
  (utils.py:475-477); results are stacked in worker order (utils.py:540-552): observations keep their dtype, reward -> float32 [B],
  done -> bool [B], the env's `info` is dropped; a reset reports zeros for both; the wrapper never resets on `done`; `seed(s)` re-seeds
  worker i with (s + i) % 2**32 and returns the workers' returns in order (utils.py:522-532)."""

  def __init__(self, g, n, seed):
    self.envs = [Scripted(g, (seed + i) % 2 ** 32) for i in range(n)]

  def _stack(self, outs):
    return ((np.stack([o[0][0] for o in outs]), np.stack([o[0][1] for o in outs])),
            np.asarray([o[1] for o in outs], np.float32), np.asarray([o[2] for o in outs], np.bool_))

  def reset(self):
    return self._stack([(e.reset(), 0.0, False) for e in self.envs])

  def step(self, actions):
    return self._stack([e.step(a) for e, a in zip(self.envs, actions)])

  def seed(self, s):
    return [e.seed((s + i) % 2 ** 32) for i, e in enumerate(self.envs)]


@pytest.mark.parametrize('tag', ['p0', 'p1'])
def test_the_statement_of_parallel_env_reproduces_the_reference_record(gold, tag):
  g = gold
  n, seed = int(g[tag + '_n']), int(g[tag + '_seed'])
  assert int(g[tag + '_batch_size']) == n and int(g[tag + '_multiprocessing']) == 1
  b = Batch(g, n, seed)
  actions = g[tag + '_actions']
  assert int(g[tag + '_n_calls']) == len(actions) + 1
  expect = [b.reset()] + [b.reset() if (a < 0).all() else b.step(a) for a in actions]
  for k, ((om, oo), r, d) in enumerate(expect):
    pre = '{}_c{}_'.format(tag, k)
    for name, mine in (('om', om), ('oo', oo), ('r', r), ('d', d)):
      ref = g[pre + name]
      assert ref.dtype == mine.dtype and ref.shape == mine.shape, (k, name, ref.dtype, ref.shape)
      assert np.array_equal(ref, mine), (k, name)
  # the wrapper does not reset on `done`: the call after the first `done` is one more step of the same episode
  L = int(g['L'])
  assert g[tag + '_c{}_d'.format(L)].all() and g[tag + '_c{}_d'.format(L + 1)].all()
  # seed(): worker i re-seeded with (s + i) mod 2^32, the list of the workers' returns comes back; the next reset shows it
  assert np.array_equal(g[tag + '_reseed'], np.asarray(b.seed(1000), np.int64))
  assert np.array_equal(g[tag + '_after_reseed_om'], b.reset()[0][0])


def test_seed_wraps_at_two_to_the_32(gold):
  g = gold
  assert int(g['p1_seed']) == 2 ** 32 - 1                     # worker 1 of that session was made with seed 0
  e = Scripted(g, 0)
  assert np.array_equal(g['p1_c0_om'][1], e.reset()[0])


def test_blocking_and_specs_of_the_record(gold):
  g = gold
  # block=None -> False: reset and step hand back a callable that receives (utils.py:326-327, :466-472, :482-488)
  assert int(g['p0_block']) == -1 and int(g['p0_reset_is_callable']) == 1 and int(g['p0_step_is_callable']) == 1
  assert int(g['p1_block']) == 1 and int(g['p1_reset_is_callable']) == 0 and int(g['p1_step_is_callable']) == 0
  # specs are those of ONE env (no batch dimension), uint8 observations, a scalar int64 action
  assert list(g['p0_obs_spec_shapes']) == ['(8, 8, 2)', '(4, 4, 1)'] and list(g['p0_obs_spec_dtypes']) == ['uint8', 'uint8']
  assert list(g['p0_action_spec']) == ['int64', '()']
  # sample(): B draws from the action space, as an int64 [B] tensor (utils.py:534-538)
  assert g['p0_samples'].shape == (3, 3) and str(g['p0_sample_dtype']) == 'int64'
  assert (g['p0_samples'] >= 0).all() and (g['p0_samples'] < int(g['A'])).all()
  # `Env`: a batch of one, same tuple layout (utils.py:283-300)
  assert int(g['e_batch_size']) == 1 and int(g['e_multiprocessing']) == 0
  assert g['e_reset_r'].dtype == np.float32 and g['e_reset_r'].shape == (1,) and g['e_reset_d'].dtype == np.bool_
  e = Scripted(g, 4)
  assert np.array_equal(g['e_reset_om'][0], e.reset()[0])
  (om, oo), r, d = e.step(7)
  assert np.array_equal(g['e_step_om'][0], om) and np.array_equal(g['e_step_oo'][0], oo)
  assert g['e_step_r'][0] == np.float32(r) and bool(g['e_step_d'][0]) == d
  assert list(g['e_sample_shape']) == [1]


# ------------------------------------------------------------------------------------------------ the product's side of B1
@pytest.mark.gpu
def test_vec_stack_env_shows_the_recorded_interface(gold):
  import torch
  from stackrl_amd import env as E
  g = gold
  n = int(g['p0_n'])
  kw = dict(episode_length=3, resolution_factor=4, sim_time_step=1 / 60.)   # maps 64 x 64 x 2 and 16 x 16 x 1
  env = E.VecStackEnv(n_parallel=n, seed=11, **kw)
  try:
    assert env.batch_size == n
    spec = env.observation_spec
    assert tuple(spec[0].shape) == (64, 64, 2) and tuple(spec[1].shape) == (16, 16, 1)      # one env's shapes, as the record's
    assert spec[0].dtype == torch.uint8 and env.action_spec.dtype == torch.int64 and tuple(env.action_spec.shape) == ()
    r = env.reset()
    assert callable(r) == bool(g['p0_reset_is_callable'])
    (om, oo), rew, done = r()
    assert om.dtype == torch.uint8 and tuple(om.shape) == (n, 64, 64, 2) and tuple(oo.shape) == (n, 16, 16, 1)
    assert rew.dtype == torch.float32 and tuple(rew.shape) == g['p0_c0_r'].shape and not rew.any()
    assert done.dtype == torch.bool and tuple(done.shape) == g['p0_c0_d'].shape and not done.any()
    a = env.sample()
    assert a.dtype == torch.int64 and tuple(a.shape) == (n,) and int(a.min()) >= 0 and int(a.max()) < env.n_actions
    s = env.step(a)
    assert callable(s) == bool(g['p0_step_is_callable'])
    (om, oo), rew, done = s()
    assert rew.dtype == torch.float32 and tuple(rew.shape) == (n,) and done.dtype == torch.bool and tuple(done.shape) == (n,)
    seeds = env.seed(2 ** 32 - 1)                        # the first item of each env's return is its seed, as the record's
    assert [x[0] for x in seeds] == [(2 ** 32 - 1 + i) % 2 ** 32 for i in range(n)]
    assert [x[0] for x in env.seed(1000)] == [int(v) for v in g['p0_reseed'][:, 0]]
    # env i is the env a batch of one makes with seed + i: row i of the batch equals that env's own reset
    (om, oo), _, _ = env.reset(block=True)
    for i in range(n):
      one = E.VecStackEnv(n_parallel=1, seed=1000 + i, **kw)
      (om1, oo1), _, _ = one.reset(block=True)
      assert torch.equal(om1[0], om[i]) and torch.equal(oo1[0], oo[i])
      one.close()
  finally:
    env.close()
  blk = E.VecStackEnv(n_parallel=2, block=True, seed=0, **kw)
  try:
    r = blk.reset()
    assert callable(r) == bool(g['p1_reset_is_callable'])
    assert callable(blk.step(blk.sample())) == bool(g['p1_step_is_callable'])
  finally:
    blk.close()
