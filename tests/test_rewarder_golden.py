"""CPU tests: the oracle's reward / goal / observation-packing / episode-machine restatement against golden vectors produced
by the reference's OWN rewarder.py, env.py and observer.py (tests/golden/make_rewarder_golden.py: the three files run as
they are over a scripted stand-in for pybullet).  Rows R1, R2, E1, E4 of SURVEY.md section 8a.  No GPU: the `-m gpu` parity
tests hold the HIP kernels to the oracle bit for bit, so they inherit this pin."""
import ast
import os

import numpy as np
import pytest

from stackrl_amd.config import StackConfig

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rewarder_golden.npz')

# Rewards: the reference sums float32 maps pairwise (numpy) and carries the metric / memory in float64 (rewarder.py:162-179);
# the oracle (and the kernels) sum in a fixed float32 tree and keep float32 throughout.  Stated tolerance on a reward of
# magnitude <= scale: 1e-6 x max(1, scale) (observed: 2e-7) — a few float32 ulps of the metric times the scale.
REWARD_TOL = 1e-6


@pytest.fixture(scope='module')
def gold():
  return np.load(GOLDEN)


def _cfg(gold, tag, **extra):
  kw = dict(ast.literal_eval(str(gold[tag + '_kwargs'])))
  kw.pop('seed'); kw.pop('dtype')
  kw.update(extra)
  return StackConfig(n_envs=1, **kw)


def _cases():
  return [str(t) for t in np.load(GOLDEN)['cases']]


@pytest.mark.parametrize('tag', _cases())
def test_observation_packing_matches_env_py(oracle_mod, gold, tag):
  """E4 (env.py:225-231, :171-172): uint8(stack([H, G]) * 255 / max(max_z, omd)) of the elevation maps the reference's
  Observer decoded from the fixture's depth buffers — bit-exact, every call incl. the reset observations."""
  cfg = _cfg(gold, tag)
  for k in range(len(gold[tag + '_action'])):
    H = oracle_mod.depth_to_elevation(cfg, 0, gold[tag + '_d_over'][k])
    O = oracle_mod.depth_to_elevation(cfg, 1, gold[tag + '_d_obj'][k])
    om, oo = oracle_mod.pack_observation(cfg, H, gold[tag + '_goal'][k], O)
    assert np.array_equal(om, gold[tag + '_obs_map'][k]), (tag, k)
    assert np.array_equal(oo, gold[tag + '_obs_obj'][k]), (tag, k)
    u, v, h, w = gold[tag + '_goal'][k]
    assert (om[u:u + h, v:v + w, 1] > 0).all() and int((om[..., 1] > 0).sum()) == h * w      # channel 1 = the goal plane


@pytest.mark.parametrize('tag', _cases())
def test_action_unflatten_and_pose_match_env_py(oracle_mod, gold, tag):
  """E1 (env.py:240-241) + E3: the pose `StackEnv.step` hands the simulator for a flat action = `Observer.pose` of
  (action // AW, action % AW) on the maps of the previous observation — the oracle's `pose`, exact."""
  cfg = _cfg(gold, tag)
  AW = cfg.overhead_res - cfg.object_res + 1
  assert int(gold[tag + '_n_actions']) == cfg.n_actions == AW * AW
  n = 0
  for k in range(1, len(gold[tag + '_action'])):
    if gold[tag + '_was_reset'][k]:
      assert np.isnan(gold[tag + '_asked'][k]).all()          # the auto-reset call places nothing (env.py:235-236)
      continue
    H = oracle_mod.depth_to_elevation(cfg, 0, gold[tag + '_d_over'][k - 1])
    O = oracle_mod.depth_to_elevation(cfg, 1, gold[tag + '_d_obj'][k - 1])
    a = int(gold[tag + '_action'][k])
    got = oracle_mod.pose(cfg, H, O, a // AW, a % AW)
    assert np.array_equal(got.astype(np.float64), gold[tag + '_asked'][k]), (tag, k, got, gold[tag + '_asked'][k])
    n += 1
  assert n >= 2


@pytest.mark.parametrize('tag', _cases())
def test_rewards_match_rewarder_py(oracle_mod, gold, tag):
  """R1 (rewarder.py:144-179, :261-307): reward = scale x (metric_t - metric_{t-1}) with the memory cleared at every reset,
  for IoU / OR / DIoU / DOR / 'all' / 'eval', integer and None exponents, scale None -> n_objects."""
  cfg = _cfg(gold, tag)
  scale = float(gold[tag + '_scale'])
  assert scale == (cfg.episode_length if cfg.reward_scale is None else cfg.reward_scale)      # rewarder.py:97
  mem = np.zeros(4, np.float32)
  worst, nonzero = 0.0, 0
  for k in range(len(gold[tag + '_action'])):
    want = gold[tag + '_reward'][k]
    if gold[tag + '_was_reset'][k]:
      mem[:] = 0                                             # rewarder.py:191-194
      assert not want.any()                                  # env.py:235-236: (reset(), 0., False, {})
      continue
    H = oracle_mod.depth_to_elevation(cfg, 0, gold[tag + '_d_over'][k])
    n = int(gold[tag + '_npos'][k])
    got = oracle_mod.rewarder_call(cfg, H, gold[tag + '_goal'][k], gold[tag + '_pos'][k][:n], gold[tag + '_dist'][k][:n], mem)
    assert got.shape == want.shape
    err = float(np.abs(got.astype(np.float64) - want).max())
    worst = max(worst, err)
    nonzero += int(np.abs(want).max() > 1e-3)
    assert err <= REWARD_TOL * max(1.0, scale), (tag, k, got, want)
  assert nonzero >= 2, 'fixture case {} exercises no reward'.format(tag)
  print(tag, 'max reward error', worst)


@pytest.mark.parametrize('tag', _cases())
def test_goal_rectangles_match_rewarder_py(oracle_mod, gold, tag):
  """R2 (rewarder.py:211-259): the goal of every episode of the case from its explicit draw list."""
  cfg = _cfg(gold, tag)
  draws = gold[tag + '_goal_draws']
  resets = np.nonzero(gold[tag + '_was_reset'])[0]
  assert len(draws) == len(resets)
  for (swap, x24, ru, rv, ba, bb), k in zip(draws, resets):
    assert (ba, bb) == (1 + 2 * swap, 3 - 2 * swap)          # rewarder.py:227-231: the bit only picks Beta(1,3) or Beta(3,1)
    assert oracle_mod.goal_from_draws(cfg, x24, ru, rv).tolist() == gold[tag + '_goal'][k].tolist(), (tag, k)


def test_goal_rectangle_table(oracle_mod, gold):
  """R2 over six geometries / area ratios x 40 draw lists each (end points of every range included): integer geometry
  exact; the goal volume (rewarder.py:258) equals h w goal_z, the OR denominator of the oracle."""
  cols = str(gold['goals_columns']).split()
  rows = gold['goals']
  assert rows.shape == (240, len(cols))
  for r in rows:
    d = dict(zip(cols, (int(x) for x in r)))
    osr = d['H'] // d['h']
    cfg = StackConfig(resolution_factor={32: 5, 16: 4}[d['h']], observable_size_ratio=osr, goal_size_ratio=d['ratio_x1000'] / 1000.0)
    assert cfg.overhead_res == d['H'] and cfg.object_res == d['h']
    got = oracle_mod.goal_from_draws(cfg, d['x24'], d['ru'], d['rv']).tolist()
    assert got == [d['u'], d['v'], d['gh'], d['gw']], d
    assert (d['beta_a'], d['beta_b']) == (1 + 2 * d['swap_bit'], 3 - 2 * d['swap_bit'])
    assert abs(d['volume_x1e6'] - d['gh'] * d['gw'] * 0.25 * 1e6) <= 1


@pytest.mark.parametrize('tag', _cases())
def test_episode_machine_matches_env_py(oracle_mod, gold, ref_pool, tag):
  """E1 (env.py:233-264, :266-293): a fresh env resets on its first step() call; `done` comes with the L-th placement; the
  call after it is the auto-reset returning (observation, 0.0, False); the rock count follows.  The oracle env (its own
  physics) against the done / reset / rock-count sequence the reference's `StackEnv` produced."""
  cfg = _cfg(gold, tag)
  env = oracle_mod.OracleEnv(cfg, ref_pool, seed=3)
  rng = np.random.RandomState(1)
  for k in range(len(gold[tag + '_action'])):
    _, r, d = env.step(rng.randint(0, cfg.n_actions, size=1).astype(np.int64))
    assert bool(d[0]) == bool(gold[tag + '_done'][k]), (tag, k)
    assert int(env.state()[1][0]) == int(gold[tag + '_npos'][k]), (tag, k)
    if gold[tag + '_was_reset'][k]:
      assert not np.asarray(r).any() and not d.any()
    assert np.asarray(r).reshape(-1).shape == gold[tag + '_reward'][k].shape
