"""CPU tests: the oracle against the golden vectors produced by the reference's own observer.py
(tests/golden/make_observer_golden.py) and against closed forms.  No GPU."""
import os

import numpy as np
import pytest

from stackrl_amd.config import StackConfig

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'observer_golden.npz')


@pytest.fixture(scope='module')
def gold():
  return np.load(GOLDEN)


CFGS = {'a': dict(resolution_factor=5), 'b': dict(resolution_factor=4)}


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_depth_to_elevation_matches_reference_bitwise(oracle_mod, gold, tag):
  cfg = StackConfig(**CFGS[tag])
  assert str(gold[tag + '_elev_over_dtype']) == 'float32'      # the reference keeps float32 (O1 quantisation)
  for k in range(gold[tag + '_depth_over'].shape[0]):
    e = oracle_mod.depth_to_elevation(cfg, 0, gold[tag + '_depth_over'][k])
    assert np.array_equal(e, gold[tag + '_elev_over'][k])
    e = oracle_mod.depth_to_elevation(cfg, 1, gold[tag + '_depth_obj'][k])
    assert np.array_equal(e, gold[tag + '_elev_obj'][k])        # includes the column flip observer.py:277


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_pose_matches_reference(oracle_mod, gold, tag):
  cfg = StackConfig(**CFGS[tag])
  for k in range(gold[tag + '_pose_pixel'].shape[0]):
    H, O = gold[tag + '_elev_over'][k], gold[tag + '_elev_obj'][k]
    for (u, v), xyz in zip(gold[tag + '_pose_pixel'][k], gold[tag + '_pose_xyz'][k]):
      got = oracle_mod.pose(cfg, H, O, int(u), int(v))
      assert np.array_equal(got.astype(np.float64), xyz), (u, v, got, xyz)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_geometry_constants(gold, tag):
  cfg = StackConfig(**CFGS[tag])
  assert gold[tag + '_shape'].tolist() == [[cfg.overhead_res] * 2, [cfg.object_res] * 2]
  assert np.allclose(gold[tag + '_size'], [cfg.overhead_res * cfg.pixel_size] * 2 + [cfg.max_z])
  assert float(gold[tag + '_max_z']) == cfg.max_z - cfg.object_max_dimension
  assert np.allclose(gold[tag + '_p2xy'], [5 * cfg.pixel_size, (77 % cfg.overhead_res) * cfg.pixel_size])
  assert gold[tag + '_xy2p'].tolist() == [0.1234 // cfg.pixel_size, 0.2345 // cfg.pixel_size]
  assert cfg.n_actions == (cfg.overhead_res - cfg.object_res + 1) ** 2


def test_render_cuboid_on_ground_is_exact_plateau(oracle_mod, ref_pool):
  """Axis-aligned `0_*` cuboid resting on the ground: plateau height = its z extent (closed form),
  on the reference's float32 elevation lattice."""
  cfg = StackConfig(n_envs=1, episode_length=4)
  env = oracle_mod.OracleEnv(cfg, ref_pool, seed=1)
  cub = [i for i, n in enumerate(ref_pool.names) if n.startswith('0_')][0]
  v, _, mc = ref_pool.mesh(cub)
  ext = v.max(0) - v.min(0)
  zc = float(ext[2]) / 2
  H = env.render_heightmap(np.array([[0.25, 0.25, zc, 0, 0, 0, 1]], np.float32), np.array([cub], np.int32))
  inside = H > 0
  n_expected = round(ext[0] / cfg.pixel_size) * round(ext[1] / cfg.pixel_size)
  assert abs(int(inside.sum()) - n_expected) <= 2 * (ext[0] + ext[1]) / cfg.pixel_size
  assert np.abs(H[inside] - ext[2]).max() < 1.2e-4          # 2x the reference's own 6.1e-5 quantum
  assert H[~inside].max() == 0.0
  # rows <-> +x, cols <-> +y (observer.py:95, :384-386): a cuboid elongated in x spans more rows than cols
  rows = np.where(inside.any(1))[0]; cols = np.where(inside.any(0))[0]
  assert (rows.max() - rows.min()) > (cols.max() - cols.min())
  assert abs((rows.min() + rows.max() + 1) / 2 - 64) <= 1 and abs((cols.min() + cols.max() + 1) / 2 - 64) <= 1


def test_object_map_is_bbox_top_minus_underside(oracle_mod, ref_pool):
  cfg = StackConfig(n_envs=1, episode_length=4)
  env = oracle_mod.OracleEnv(cfg, ref_pool, seed=1)
  cub = [i for i, n in enumerate(ref_pool.names) if n.startswith('0_')][0]
  v, _, _ = ref_pool.mesh(cub)
  O = env.render_object(cub)
  hit = O > 1e-4
  # O = (z_c + oz/2) - z_underside with z_underside = -ext_z/2 in the link frame
  expect = cfg.object_max_dimension / 2 + (v[:, 2].max() - v[:, 2].min()) / 2
  assert np.abs(O[hit] - expect).max() < 1.2e-4
  assert O[~hit].max() == 0.0
  # pose(): a rock dropped on empty ground rests with its underside on z = 0
  H = np.zeros((cfg.overhead_res,) * 2, np.float32)
  x, y, z = oracle_mod.pose(cfg, H, O, 10, 20)
  assert abs(z - (v[:, 2].max() - v[:, 2].min()) / 2) < 1.2e-4
  assert x == 10 * cfg.pixel_size + 0.0625 and y == 20 * cfg.pixel_size + 0.0625


def test_iou_sums_against_numpy(oracle_mod):
  cfg = StackConfig()
  rng = np.random.RandomState(0)
  H = (rng.uniform(0, 0.3, size=(128, 128)) * (rng.uniform(size=(128, 128)) > 0.6)).astype(np.float32)
  rect = np.array([20, 30, 64, 50], np.int32)
  G = np.zeros_like(H); G[20:84, 30:80] = 0.25
  inter, uni = oracle_mod.iou_sums(cfg, H, rect)
  ref_i = np.sum(np.minimum(H[G != 0], 0.25))       # rewarder.py:297-301
  ref_u = np.sum(np.maximum(H, G))                  # rewarder.py:303-307
  assert abs(inter - ref_i) <= 2e-6 * ref_i and abs(uni - ref_u) <= 2e-6 * ref_u


def test_uint8_packing_truncates(oracle_mod, ref_pool):
  """env.py:171-172: np.array(x*255/0.375, dtype=uint8) truncates; goal plane packs to exactly 170."""
  cfg = StackConfig(n_envs=2, episode_length=4)
  env = oracle_mod.OracleEnv(cfg, ref_pool, seed=3)
  (om, oo), r, d = env.reset()
  Hm, Om, g = env.maps()
  exp = (Hm * np.float32(255) / np.float32(0.375)).astype(np.uint8)
  assert np.array_equal(om[..., 0], exp)
  assert set(np.unique(om[..., 1]).tolist()) == {0, 170}
  for i in range(2):
    u, v, h, w = g[i]
    assert (om[i, u:u + h, v:v + w, 1] == 170).all() and int((om[i, ..., 1] == 170).sum()) == h * w
  assert np.array_equal(oo[..., 0], (Om * np.float32(255) / np.float32(0.375)).astype(np.uint8))
  assert r.tolist() == [0, 0] and not d.any()


def test_goal_rectangle_ranges(oracle_mod):
  """rewarder.py:225-253: h in [32,128], w = clamp(4096//h, 32, 128), offsets inside the 1/8 margins."""
  cfg = StackConfig()
  hs = []
  for key in range(300):
    u, v, h, w = oracle_mod.goal_from_rng(cfg, key, 1)
    assert 32 <= h <= 128 and w == min(max(32, 4096 // h), 128)
    assert (128 - h) // 8 <= u <= 7 * (128 - h) // 8 and (128 - w) // 8 <= v <= 7 * (128 - w) // 8
    hs.append(h)
  hs = np.array(hs)
  assert (hs < 56).mean() > 0.2 and (hs > 104).mean() > 0.2 and ((hs > 70) & (hs < 90)).mean() < 0.2   # Beta(1,3)/Beta(3,1) mixture is bimodal


def test_acos_polynomial(oracle_mod):
  L = oracle_mod.lib()
  xs = np.linspace(-1, 1, 2001)
  got = np.array([L.srlo_acosf(float(x)) for x in xs])
  assert np.abs(got - np.arccos(xs)).max() < 1e-6


def test_episode_state_machine(oracle_mod, ref_pool):
  """env.py:233-247: L placements, done on the L-th, next call is the auto-reset (obs, 0.0, False)."""
  L = 5
  cfg = StackConfig(n_envs=3, episode_length=L)
  env = oracle_mod.OracleEnv(cfg, ref_pool, seed=7)
  # first step() on a fresh env resets (env.py:219-220, :235-236)
  (om, oo), r, d = env.step(np.zeros(3, np.int64))
  assert r.tolist() == [0, 0, 0] and not d.any() and env.state()[1].tolist() == [0, 0, 0]
  assert oo.max() > 0                                  # first rock pending at the spawn pose
  for k in range(L):
    (om, oo), r, d = env.step(env.sample())
    assert env.state()[1].tolist() == [k + 1] * 3
    assert d.tolist() == [k == L - 1] * 3
  assert oo.max() == 0                                 # terminal observation: nothing pending (observer.py:262-277)
  (om, oo), r, d = env.step(env.sample())
  assert r.tolist() == [0, 0, 0] and not d.any() and env.state()[1].tolist() == [0, 0, 0]
  # mesh ids are drawn without replacement (env.py:268-272)
  for _ in range(L):
    env.step(env.sample())
  ids = env.state()[0][:, :L, 7].astype(int)
  for row in ids:
    assert len(set(row.tolist())) == L


def test_physics_invariants(oracle_mod, ref_pool):
  """Tier C invariants of the settle solver: rest speed below threshold, rocks above ground,
  bounded penetration, nothing diverged."""
  n, L = 64, 12
  cfg = StackConfig(n_envs=n, episode_length=L)
  env = oracle_mod.OracleEnv(cfg, ref_pool, seed=11)
  env.reset()
  pens = []
  for k in range(L):
    env.step(env.sample())
    poses, nb, sub, st = env.state()
    v = env.velocities()
    speed = np.linalg.norm(v[..., :3], axis=-1)
    assert speed.max() <= cfg.velocity_threshold + 1e-7          # simulator.py:328-335
    assert (st == 0).all()
    assert (sub.sum(1) < 3000).all()
    mp, npts = env.contacts()
    pens.append(mp)
    assert (poses[:, :k + 1, 2] > 0.005).all() and (poses[:, :k + 1, 2] < 0.375).all()
    q = poses[:, :k + 1, 3:7]
    assert np.abs(np.linalg.norm(q, axis=-1) - 1).max() < 1e-5
  # Penetration left when the stop criterion fires (simulator.py:322-335 looks at linear speeds only).  A contact at
  # rest is pushed out at erp x depth / dt, so a depth above velocity_threshold x dt / erp = 0.5 mm keeps its bodies
  # moving: the bulk sits below 1 mm.  The tail is not a solver residue (it is the same with 10 or 50 sweeps): `Observer.pose`
  # releases a rock by pixel-centre samples of both maps (observer.py:405-413), which on sloped faces starts it up to
  # half a pixel x slope inside its neighbours, and a rock wedged between two others cannot be pushed out of both.
  pens = np.stack(pens)
  assert np.quantile(pens, 0.95) <= 1e-3
  assert (pens > 1e-3).mean() <= 0.02
  assert pens.max() < 0.006


def test_all_and_eval_metrics(oracle_mod, ref_pool):
  """`Rewarder` with metric 'all' / 'eval' (rewarder.py:147-160): 'all' returns the four metrics' rewards at once — each
  column equals the reward of a run with that single metric on the same script; 'eval' returns the IoU reward and 'AD',
  the change of the average discount of ALL rocks (unscaled): with a goal that covers the whole map every rock is inside
  it, so the running average discount times the number of rocks equals the running DOR times the episode length."""
  n, L = 3, 6
  rng = np.random.RandomState(2)
  ids = np.stack([rng.choice(len(ref_pool), size=L, replace=False) for _ in range(n)]).astype(np.int32)
  rect = np.array([[0, 0, 128, 128]] * n, np.int32)
  acts = rng.randint(0, 9409, size=(L, n)).astype(np.int64)

  def run(metric, scale=2.0):
    env = oracle_mod.OracleEnv(StackConfig(n_envs=n, episode_length=L, rewarder=metric, reward_scale=scale), ref_pool, seed=3)
    env.set_script(ids, rect)
    _, r0, _ = env.reset()
    out = [env.step(acts[k])[1] for k in range(L)]
    _, rr, _ = env.step(acts[0])                     # the auto-reset call: zeros in every column
    assert not rr.any() and r0.shape == rr.shape
    return np.stack(out)
  single = {m: run(m) for m in ('iou', 'or', 'diou', 'dor')}
  allr = run('all')
  assert allr.shape == (L, n, 4) and StackConfig(rewarder='all').reward_keys == ('IoU', 'OR', 'DIoU', 'DOR')
  for col, m in enumerate(('iou', 'or', 'diou', 'dor')):
    assert np.array_equal(allr[..., col], single[m]), m
  ev = run('eval')
  assert ev.shape == (L, n, 2) and StackConfig(rewarder='eval').reward_keys == ('IoU', 'AD')
  assert np.array_equal(ev[..., 0], single['iou'])
  ad = np.cumsum(ev[..., 1], axis=0)                 # running average discount (not scaled)
  dor = np.cumsum(single['dor'], axis=0) / 2.0       # running DOR (undo the scale)
  k = np.arange(1, L + 1, dtype=np.float64)[:, None]
  np.testing.assert_allclose(ad * k, dor * L, rtol=0, atol=2e-5)
