"""Heuristic baseline policies: oracle vs the reference's own golden vectors on CPU,
HIP kernels vs both on the GPU."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'baselines_golden.npz')
CASES = [('height', 'height', {}), ('difference', 'difference', {}),
         ('difference_e1w0', 'difference', dict(difference_exponent=1, weights_exponent=0)),
         ('corrcoef', 'corrcoef', {}), ('corrcoef_localized', 'corrcoef', dict(localized=True)),
         ('correlate', 'correlate', {})]


@pytest.fixture(scope='module')
def gold():
  return np.load(GOLD)


def test_oracle_matches_reference_golden(gold):
  from oracle import baselines_oracle as B
  for k in range(gold['obs_map'].shape[0]):
    inp = (gold['obs_map'][k], gold['obs_obj'][k])
    for name, fn, kw in CASES:
      got = B.METHODS[fn](inp, **kw)
      assert np.abs(got - gold[name][k]).max() <= 1e-12 * max(1e-300, np.abs(gold[name][k]).max()), (name, k)
    assert np.array_equal(B.goal_overlap(inp), gold['goal_overlap'][k])
    for method in ('height', 'difference', 'corrcoef', 'correlate'):
      for goal, mo in ((True, 1), (True, 0), (False, 1)):
        tag = 'select_{}_g{}_m{}'.format(method, int(goal), mo)
        a, v = B.select(gold[method][k], gold['goal_overlap'][k], goal, mo)
        assert a == gold[tag + '_action'][k], (tag, k)
        assert np.allclose(v, gold[tag + '_values'][k], rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_hip_value_maps_match_reference(gold):
  torch = pytest.importorskip('torch')
  from stackrl_amd import baselines as Bd
  inp = (torch.from_numpy(gold['obs_map']).cuda(), torch.from_numpy(gold['obs_obj']).cuda())
  for name, fn, kw in CASES:
    vals, mask = Bd.heuristic_values(fn, inp, **kw)
    ref = gold[name]
    if fn == 'height':
      assert np.array_equal(vals.cpu().numpy(), ref)                       # max of exact sums: bit-exact
    else:                                                                 # float64 sums in a different order: 1e-12
      assert np.abs(vals.cpu().numpy() - ref).max() <= 1e-12 * max(1e-300, np.abs(ref).max()), name
    assert np.array_equal(mask.cpu().numpy(), gold['goal_overlap'])       # integer arithmetic: exact


@pytest.mark.gpu
def test_hip_selection_matches_reference(gold):
  torch = pytest.importorskip('torch')
  from stackrl_amd import baselines as Bd
  mask = torch.from_numpy(gold['goal_overlap']).cuda()
  for method in ('height', 'difference', 'corrcoef', 'correlate'):
    vals = torch.from_numpy(gold[method]).cuda()
    for goal, mo in ((True, 1), (True, 0), (False, 1)):
      tag = 'select_{}_g{}_m{}'.format(method, int(goal), mo)
      a, neg = Bd.select(vals, mask, goal=goal, minorder=mo, value=True)
      assert np.array_equal(a.cpu().numpy(), gold[tag + '_action']), tag   # placement indices bit-exact
      assert np.array_equal(neg.cpu().numpy(), gold[tag + '_values']), tag


@pytest.mark.gpu
def test_baseline_policy_end_to_end(gold, ref_pool):
  """The policy object on live env observations agrees with the oracle on the same observations."""
  torch = pytest.importorskip('torch')
  from oracle import baselines_oracle as B
  from stackrl_amd import baselines as Bd, env as envs
  env = envs.make('Stack-v0', n_parallel=6, seed=2, pool=ref_pool, episode_length=5, block=True)
  obs, _, _ = env.reset()
  for _ in range(3):
    obs, _, _ = env.step(env.sample())
  for method in ('height', 'difference', 'corrcoef', 'correlate'):
    pol = Bd.Baseline(method=method, goal=True, minorder=1)
    acts = pol(obs).cpu().numpy()
    om, oo = obs[0].cpu().numpy(), obs[1].cpu().numpy()
    for i in range(6):
      inp = (om[i], oo[i])
      vals = B.METHODS[method](inp)
      a, _ = B.select(vals, B.goal_overlap(inp), True, 1)
      if a != acts[i]:     # float64 sums differ in the last bits between numpy and the kernel: accept exact ties only
        assert abs(vals.flat[a] - vals.flat[acts[i]]) <= 1e-12 * max(1.0, abs(vals.flat[a])), (method, i)
  with pytest.raises(ValueError):
    Bd.Baseline(method='nope')                       # baselines.py:184-187
  env.close()
