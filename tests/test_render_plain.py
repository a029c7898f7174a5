"""The renderer's culled definition (oracle `render_heightmap`: item rows, list ranges, column spans — what the kernels
restate) held to THE PLAIN STATEMENT of the overhead map (`srlo_render_heightmap_all`: every pixel of a rock's bounding box
against all up-facing faces and all outline sides; and the hull-interval form over all planes, which uses no edge table).

Reference: one `getCameraImage` call of the overhead camera, observer.py:252-260.  The plain function is never touched by a
render optimisation; any later culling has to keep these tests green.

Bounds:
  * after the depth codec (what Observer.state[0] / the observation hold): exactly equal;
  * before the codec: equal up to the last bit, and only where two faces are coplanar within rounding (the cuboids' split
    quads, near-flat facets) — asserted as <= 1 ulp of the plain value, and the differing pixels counted;
  * coverage (which pixels see a rock at all): identical.
The `-m gpu` half holds `srl_render_heightmap` (HIP, through the C-ABI) to the plain statement under the same bounds.
"""
import numpy as np
import pytest

from stackrl_amd import assets
from stackrl_amd.config import StackConfig


def _scenes(rng, n, pool_size, max_rocks=12, tilt='any'):
  """n random scenes: (poses [k, 7], mesh ids [k]); rocks anywhere over the map (partly outside it too), any orientation."""
  out = []
  for _ in range(n):
    k = rng.randint(1, max_rocks + 1)
    poses = np.zeros((k, 7), np.float32)
    for b in range(k):
      if tilt == 'flat':            # resting orientations: yaw only (+ a flip), where coplanar top faces are the rule
        a = rng.uniform(0, 2 * np.pi)
        q = np.array([0, 0, np.sin(a / 2), np.cos(a / 2)])
        if rng.randint(2):
          q = np.array([np.cos(a / 2), np.sin(a / 2), 0, 0])
      else:
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
      poses[b] = [rng.uniform(-0.03, 0.53), rng.uniform(-0.03, 0.53), rng.uniform(0.0, 0.3), *q]
    out.append((poses, rng.randint(pool_size, size=k).astype(np.int32)))
  return out


def _ulp_diff(a, b):
  ia = a.view(np.int32).astype(np.int64)
  ib = b.view(np.int32).astype(np.int64)
  return np.abs(ia - ib)


def _check(o, scenes, tag):
  covered = differing_raw = hull_differs = 0
  for s, (poses, mesh) in enumerate(scenes):
    culled = o.render_heightmap(poses, mesh)                       # the function srlo_step / srlo_reset run
    plain = o.render_heightmap_all(poses, mesh, form=1)
    hull = o.render_heightmap_all(poses, mesh, form=2)
    assert np.array_equal(culled, o.render_heightmap_all(poses, mesh, form=0)), 'form 0 is not the culled definition'
    assert np.array_equal(culled, plain), '{} scene {}: culled != plain after the codec at {} pixels'.format(
      tag, s, int((culled != plain).sum()))
    c_raw = o.render_heightmap_all(poses, mesh, form=0, raw=True)
    p_raw = o.render_heightmap_all(poses, mesh, form=1, raw=True)
    assert np.array_equal(c_raw > 0, p_raw > 0), '{} scene {}: coverage differs'.format(tag, s)
    d = _ulp_diff(c_raw, p_raw)
    assert d.max() <= 1, '{} scene {}: {} ulp before the codec'.format(tag, s, int(d.max()))
    covered += int((p_raw > 0).sum())
    differing_raw += int((d > 0).sum())
    # the hull-interval form (no edge table) agrees with the outline form except at pixel centres ON a rock's outline
    # (vertical faces of rocks lying flat: the centre may fall either way, and then shows the rock or what lies below it)
    hull_differs += int((hull != plain).sum())
    assert int((hull != plain).sum()) <= 6, '{} scene {}: outline vs hull-interval form differ at {} pixels'.format(
      tag, s, int((hull != plain).sum()))
  assert hull_differs <= covered // 20000 + 6, hull_differs
  return covered, differing_raw


@pytest.mark.parametrize('res_factor', [5, 4])        # 128 x 128 / 32 x 32 maps and configs[4]'s 64 x 64 / 16 x 16
def test_culled_definition_equals_the_plain_statement_reference_rocks(ref_pool, oracle_mod, res_factor):
  cfg = StackConfig(n_envs=1, episode_length=8, resolution_factor=res_factor)
  o = oracle_mod.OracleEnv(cfg, ref_pool, seed=1)
  rng = np.random.RandomState(50 + res_factor)
  scenes = _scenes(rng, 300, len(ref_pool)) + _scenes(rng, 100, len(ref_pool), tilt='flat')
  # every scene of the second batch gets one of the five cuboids (coplanar split quads: the documented last-bit case)
  cub = [i for i, nm in enumerate(ref_pool.names) if str(nm).startswith('0_')]
  assert len(cub) == 5
  for k, (poses, mesh) in enumerate(scenes[300:]):
    mesh[0] = cub[k % 5]
  covered, diff = _check(o, scenes, 'ref rocks {}'.format(cfg.overhead_res))
  assert covered > 100000 // (1 if res_factor == 5 else 4)
  print('reference rocks at {}: {} covered pixels, {} differ in the last bit before the codec'.format(cfg.overhead_res, covered, diff))


@pytest.mark.parametrize('res_factor', [5, 4])
def test_culled_definition_equals_the_plain_statement_pool(oracle_mod, res_factor):
  pool = assets.default_pool()            # the 5,000 synthetic rocks bench.py runs on (generator seed 11, cached)
  cfg = StackConfig(n_envs=1, episode_length=8, resolution_factor=res_factor)
  o = oracle_mod.OracleEnv(cfg, pool, seed=1)
  rng = np.random.RandomState(70 + res_factor)
  scenes = _scenes(rng, 250, len(pool)) + _scenes(rng, 50, len(pool), tilt='flat')
  covered, diff = _check(o, scenes, 'pool {}'.format(cfg.overhead_res))
  print('pool at {}: {} covered pixels, {} differ in the last bit before the codec'.format(cfg.overhead_res, covered, diff))


def test_plain_statement_on_a_cuboid(ref_pool, oracle_mod):
  """The plain function itself against the closed form: an axis-aligned cuboid flat on the ground is a plateau of its height
  over exactly the pixel centres inside its footprint (both plain forms, before and after the codec)."""
  cfg = StackConfig(n_envs=1, episode_length=8)
  o = oracle_mod.OracleEnv(cfg, ref_pool, seed=1)
  for cub in [i for i, nm in enumerate(ref_pool.names) if str(nm).startswith('0_')]:
    v, _, mc = ref_pool.mesh(cub)
    ext = (v.max(0) - v.min(0)).astype(np.float64)
    lo = v.min(0).astype(np.float64) - np.asarray(mc[1:4], np.float64)      # the cuboid's corner relative to its centre of mass
    cx, cy = 0.2512, 0.2487                                                  # off the pixel lattice: no centre on the border
    pose = np.array([[cx, cy, -lo[2], 0, 0, 0, 1]], np.float32)
    c = (np.arange(cfg.overhead_res) + 0.5) * cfg.pixel_size
    want = np.outer((c >= cx + lo[0]) & (c <= cx + lo[0] + ext[0]), (c >= cy + lo[1]) & (c <= cy + lo[1] + ext[1]))
    for form in (1, 2):
      H = o.render_heightmap_all(pose, np.array([cub], np.int32), form=form, raw=True)
      assert np.array_equal(H > 0, want), 'cuboid {} form {}: footprint'.format(cub, form)
      np.testing.assert_allclose(H[want], ext[2], rtol=0, atol=2e-6)
      Hc = o.render_heightmap_all(pose, np.array([cub], np.int32), form=form)
      assert np.array_equal(Hc > 0, want) and np.abs(Hc[want] - ext[2]).max() < 1.2e-4


@pytest.mark.gpu
@pytest.mark.parametrize('res_factor', [5, 4])
def test_hip_renderer_equals_the_plain_statement(ref_pool, oracle_mod, res_factor):
  torch = pytest.importorskip('torch')
  from stackrl_amd import env as envs
  n = 256
  for pool, tag in ((ref_pool, 'ref'), (assets.default_pool(), 'pool')):
    g = envs.VecStackEnv(n_parallel=n, seed=3, pool=pool, block=True, episode_length=8, resolution_factor=res_factor)
    o = oracle_mod.OracleEnv(StackConfig(n_envs=1, episode_length=8, resolution_factor=res_factor), pool, seed=3)
    rng = np.random.RandomState(90 + res_factor)
    scenes = _scenes(rng, n - 32, len(pool)) + _scenes(rng, 32, len(pool), tilt='flat')
    poses = np.zeros((n, 32, 7), np.float32)
    mesh = np.zeros((n, 32), np.int32)
    nb = np.zeros(n, np.int32)
    for i, (p, m) in enumerate(scenes):
      nb[i] = len(m); poses[i, :len(m)] = p; mesh[i, :len(m)] = m
    out = g.render_heightmap(torch.from_numpy(poses).cuda(), torch.from_numpy(mesh).cuda(), torch.from_numpy(nb).cuda()).cpu().numpy()
    for i, (p, m) in enumerate(scenes):
      plain = o.render_heightmap_all(p, m, form=1)
      assert np.array_equal(out[i], plain), '{} scene {}: HIP != plain statement at {} pixels (max {})'.format(
        tag, i, int((out[i] != plain).sum()), float(np.abs(out[i] - plain).max()))
    g.close()
