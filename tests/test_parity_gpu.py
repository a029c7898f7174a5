"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.

Bar: integer / index / byte outputs bit-exact; float state compared with atol 1e-6 m (the solver
definition is evaluated with identical IEEE operations on both sides, so the observed difference is 0;
the stated tolerance of the contract is 1e-4 m / 1e-3 rad, SURVEY.md section 8c tier C).
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

POSE_ATOL = 1e-6


def _mk(ref_pool, oracle_mod, n, L, seed=11, **kw):
  from stackrl_amd import env as envs
  from stackrl_amd.config import StackConfig
  g = envs.VecStackEnv(n_parallel=n, seed=seed, pool=ref_pool, block=True, episode_length=L, **kw)
  o = oracle_mod.OracleEnv(StackConfig(n_envs=n, episode_length=L, **kw), ref_pool, seed=seed)
  return g, o


def _cmp_step(g, o, gout, oout, tag):
  (gm, go), gr, gd = gout
  (om, oo), orr, od = oout
  assert np.array_equal(gm.cpu().numpy(), om), tag + ': obs_map differs'
  assert np.array_equal(go.cpu().numpy(), oo), tag + ': obs_obj differs'
  assert np.array_equal(gd.cpu().numpy(), od), tag + ': done differs'
  np.testing.assert_allclose(gr.cpu().numpy(), orr, rtol=0, atol=1e-7, err_msg=tag + ': reward')
  gp, gnb, gsub, gst = g.state()
  op, onb, osub, ost = o.state()
  assert np.array_equal(gnb, onb), tag
  assert np.array_equal(gsub, osub), tag + ': sub-step counts differ {} vs {}'.format(gsub[:4], osub[:4])
  assert np.array_equal(gst, ost), tag
  np.testing.assert_allclose(gp, op, rtol=0, atol=POSE_ATOL, err_msg=tag + ': poses')
  gH, gO, gg = g.maps()
  oH, oO, og = o.maps()
  assert np.array_equal(gg, og), tag + ': goal rect'
  np.testing.assert_allclose(gH, oH, rtol=0, atol=1.2e-4, err_msg=tag + ': height map')
  np.testing.assert_allclose(gO, oO, rtol=0, atol=1.2e-4, err_msg=tag + ': object map')
  return float(np.abs(gp - op).max()), float(np.abs(gH - oH).max())


def test_object_maps_bit_exact(ref_pool, oracle_mod):
  g, o = _mk(ref_pool, oracle_mod, 1, 8)
  for m in range(len(ref_pool)):
    assert np.array_equal(g.object_map(m), o.render_object(m)), 'mesh {}'.format(m)


def test_render_explicit_poses(ref_pool, oracle_mod):
  n = 16
  g, o = _mk(ref_pool, oracle_mod, n, 8)
  rng = np.random.RandomState(3)
  poses = np.zeros((n, 32, 7), np.float32)
  mesh = np.zeros((n, 32), np.int32)
  nb = rng.randint(0, 9, size=n).astype(np.int32)
  nb[0] = 0
  for i in range(n):
    for b in range(nb[i]):
      q = rng.normal(size=4); q /= np.linalg.norm(q)
      poses[i, b] = [rng.uniform(-0.02, 0.52), rng.uniform(-0.02, 0.52), rng.uniform(0.0, 0.3), *q]
      mesh[i, b] = rng.randint(len(ref_pool))
  out = g.render_heightmap(torch.from_numpy(poses).cuda(), torch.from_numpy(mesh).cuda(),
                           torch.from_numpy(nb).cuda()).cpu().numpy()
  for i in range(n):
    ref = o.render_heightmap(poses[i, :nb[i]], mesh[i, :nb[i]])
    assert np.array_equal(out[i], ref), 'env {} max diff {}'.format(i, np.abs(out[i] - ref).max())


@pytest.mark.parametrize('L,n,kw', [
  (8, 32, {}),
  (8, 8, dict(sim_time_step=0.0125, rewarder='dor', reward_scale=None)),   # root config.gin env overrides
  (16, 8, dict(rewarder='diou')),
  (4, 4, dict(rewarder='or', smooth_placing=False)),
  (6, 6, dict(rewarder='all', reward_scale=2.0)),      # rewarder.py:157-158: the four metrics at once, reward [B, 4]
  (6, 6, dict(rewarder='eval')),                       # rewarder.py:147-156: IoU reward + average-discount change, [B, 2]
])
def test_scripted_episodes(ref_pool, oracle_mod, L, n, kw):
  g, o = _mk(ref_pool, oracle_mod, n, L, **kw)
  rng = np.random.RandomState(5)
  ids = np.stack([rng.choice(len(ref_pool), size=L, replace=False) for _ in range(n)]).astype(np.int32)
  rect = np.stack([[rng.randint(8, 40), rng.randint(8, 40), 64, 64] for _ in range(n)]).astype(np.int32)
  g.set_script(ids, rect); o.set_script(ids, rect)
  gout, oout = g.reset(), o.reset()
  assert np.array_equal(gout[0][0].cpu().numpy(), oout[0][0])
  assert np.array_equal(gout[0][1].cpu().numpy(), oout[0][1])
  A = g.n_actions
  worst = 0.0
  for k in range(L + 2):       # runs through done and the auto-reset step (env.py:235-236)
    a = rng.randint(0, A, size=n).astype(np.int64)
    gout = g.step(torch.from_numpy(a).cuda())
    oout = o.step(a)
    dp, dh = _cmp_step(g, o, gout, oout, 'step {}'.format(k))
    worst = max(worst, dp)
    if k == L - 1:
      assert oout[2].all() and gout[2].all()
    if k == L:
      assert not oout[2].any() and float(gout[1].abs().max()) == 0.0
    assert tuple(gout[1].shape) == tuple(oout[1].shape)
  print('max pose diff', worst)


def test_rng_driven_episodes_and_sample(ref_pool, oracle_mod):
  n, L = 64, 8
  g, o = _mk(ref_pool, oracle_mod, n, L, seed=123)
  gout, oout = g.reset(), o.reset()
  assert np.array_equal(gout[0][0].cpu().numpy(), oout[0][0])
  for k in range(2 * (L + 1)):
    ga = g.sample()
    oa = o.sample()
    assert np.array_equal(ga.cpu().numpy(), oa)
    _cmp_step(g, o, g.step(ga), o.step(oa), 'step {}'.format(k))


def test_invalid_action_raises(ref_pool, oracle_mod):
  g, o = _mk(ref_pool, oracle_mod, 4, 4)
  g.reset()
  a = torch.tensor([0, 5, g.n_actions, 7], dtype=torch.int64).cuda()
  with pytest.raises(AssertionError, match='Invalid action'):
    g.step(a)
  # state of the offending env is untouched, the others stepped
  _, nb, _, st = g.state()
  assert nb.tolist() == [1, 1, 0, 1] and st[2] & 4


def test_nonblocking_step_returns_callable(ref_pool, oracle_mod):
  from stackrl_amd import env as envs
  g = envs.make('Stack-v0', n_parallel=8, seed=1, pool=ref_pool, episode_length=4)
  step = g.reset()
  assert callable(step)
  (om, oo), r, d = step()
  assert om.shape == (8, 128, 128, 2) and oo.shape == (8, 32, 32, 1) and om.dtype == torch.uint8
  assert r.dtype == torch.float32 and d.dtype == torch.bool and not d.any()
  nxt = g.step(g.sample())
  assert callable(nxt)
  (om, oo), r, d = nxt()
  assert g.batch_size == 8 and not g.multiprocessing
  assert g.observation_spec[0].shape == (128, 128, 2) and g.action_spec.dtype == torch.int64


@pytest.mark.parametrize('L,n,kw', [
  (32, 4, {}),                                   # BASELINE config 5 episode length: 192 manifold slots, 3 points/thread
  (12, 6, dict(resolution_factor=4)),            # 64 x 64 height map, 16 x 16 object map, 2,401 actions (config 5)
  (32, 3, dict(resolution_factor=4)),            # configs[4]'s per-GPU workload together: 32 rocks (`_t512`) WITH 64 x 64 maps
  (6, 5, dict(observable_size_ratio=3)),         # 96 x 96 height map: 4.5 epilogue rounds whose groups change columns
  (5, 7, dict(resolution_factor=3)),             # 32 x 32 height map (1,024 pixels < 4 x 512 threads): the general walk with its bound on
                                                 # the group index (round 3: the aligned walk wrote past such a map), 8 x 8 object map, 625 actions
])
def test_large_configs(ref_pool, oracle_mod, L, n, kw):
  g, o = _mk(ref_pool, oracle_mod, n, L, seed=31, **kw)
  gout, oout = g.reset(), o.reset()
  assert np.array_equal(gout[0][0].cpu().numpy(), oout[0][0])
  for k in range(L + 1):
    ga, oa = g.sample(), o.sample()
    assert np.array_equal(ga.cpu().numpy(), oa)
    _cmp_step(g, o, g.step(ga), o.step(oa), 'step {}'.format(k))
  assert g.n_actions == {('resolution_factor', 4): 2401, ('observable_size_ratio', 3): 4225, ('resolution_factor', 3): 625}.get(next(iter(kw.items()), None), 9409)


def test_two_wave_settle_variant_matches_the_oracle(ref_pool, oracle_mod, monkeypatch):
  """9 - 16 rocks in batches of >= 3,072 envs run `srl_k_step_t128` (two waves per env, two contact points per thread, no
  LDS vertex copy); forced here on a small batch (SRL_STEP_VARIANT, read at srl_create) and held bit-exact against the
  oracle over a 14-rock episode and its auto-reset.  (`test_full_size_dqn_config_properties` compares it at 4,096 envs with
  the four-wave variant of a 64-env batch.)"""
  monkeypatch.setenv('SRL_STEP_VARIANT', 'two_wave')
  L, n = 14, 5
  g, o = _mk(ref_pool, oracle_mod, n, L, seed=17)
  gout, oout = g.reset(), o.reset()
  assert np.array_equal(gout[0][0].cpu().numpy(), oout[0][0])
  for k in range(L + 1):
    ga, oa = g.sample(), o.sample()
    assert np.array_equal(ga.cpu().numpy(), oa)
    _cmp_step(g, o, g.step(ga), o.step(oa), 'step {}'.format(k))


@pytest.mark.parametrize('L,n,kw', [(8, 37, {}), (12, 70, dict(resolution_factor=4))])
def test_ordered_launch_matches_the_oracle(ref_pool, oracle_mod, L, n, kw):
  """`srl_set_launch_order`: batches of >= 2,048 envs hand the settle kernel's workgroups the envs with the highest release
  first (`srl_k_order_keys` / `srl_k_order_sort`, settle.hip).  Forced here on small batches whose size is not a power of
  two (padding of the sorting network) and held bit-exact against the oracle through an episode, its `done` step and the
  auto-reset call: envs are independent, so the permutation must not show in any result."""
  from stackrl_amd import env as envs
  from stackrl_amd.config import StackConfig
  g = envs.VecStackEnv(n_parallel=n, seed=29, pool=ref_pool, block=True, episode_length=L, launch_order=True, **kw)
  o = oracle_mod.OracleEnv(StackConfig(n_envs=n, episode_length=L, **kw), ref_pool, seed=29)
  gout, oout = g.reset(), o.reset()
  assert np.array_equal(gout[0][0].cpu().numpy(), oout[0][0])
  for k in range(L + 3):
    ga, oa = g.sample(), o.sample()
    assert np.array_equal(ga.cpu().numpy(), oa)
    _cmp_step(g, o, g.step(ga), o.step(oa), 'step {}'.format(k))
  g.close()


@pytest.mark.parametrize('L,n,kw', [(8, 24, {}), (14, 6, {}), (20, 4, dict(resolution_factor=4))])
def test_tail_staged_records_equal_the_staging_kernels(ref_pool, L, n, kw):
  """The rocks' render records (csrc/stage.h) are made in the tail of the settle kernels (all three thread / point variants
  here: 128 threads, 256 threads, 256 threads with two points per thread); `srl_k_stage` makes them for a handle whose
  records are stale.  Two handles with the same seed, one of them told before every step that its bodies were moved
  (`set_body_state` with nothing to set: the records are staged again by the kernel after the settle kernel): the records of
  every placed rock, the observations and the rewards are equal bit for bit."""
  from stackrl_amd import env as envs
  a = envs.VecStackEnv(n_parallel=n, seed=13, pool=ref_pool, block=True, episode_length=L, **kw)
  b = envs.VecStackEnv(n_parallel=n, seed=13, pool=ref_pool, block=True, episode_length=L, **kw)
  a.reset(); b.reset()
  for k in range(L + 2):
    act = a.sample()
    assert torch.equal(act, b.sample())
    b.set_body_state()                       # marks b's records stale: srl_k_stage runs in its next step
    (am, ao), ar, ad = a.step(act)
    (bm, bo), br, bd = b.step(act)
    assert torch.equal(am, bm) and torch.equal(ao, bo) and torch.equal(ar, br) and torch.equal(ad, bd), 'call {}'.format(k)
    ra, rb = a.stage_records().view(np.int32), b.stage_records().view(np.int32)
    nb = a.state()[1]
    for e in range(n):
      for r in range(int(nb[e])):
        hdr = ra[e, r, :2]
        nup, nsil = int(hdr[0, 2]), int(hdr[0, 3])
        used = 10 + nup + nsil
        assert np.array_equal(ra[e, r, :2], rb[e, r, :2]), 'call {} env {} rock {}: header'.format(k, e, r)
        if int(hdr[1, 0]) > 0:               # (a rock outside the window has no tables)
          assert np.array_equal(ra[e, r, 2:10], rb[e, r, 2:10]), 'call {} env {} rock {}: span / range tables'.format(k, e, r)
          # the two lists are SETS per first item row (filled with LDS counters): compare them sorted
          pa, pb_ = ra[e, r, 10:10 + nup, :3], rb[e, r, 10:10 + nup, :3]
          sa, sb = ra[e, r, 10 + nup:used, :3], rb[e, r, 10 + nup:used, :3]
          for x, y, what in ((pa, pb_, 'planes'), (sa, sb, 'sides')):
            assert np.array_equal(x[np.lexsort(x.T)], y[np.lexsort(y.T)]), 'call {} env {} rock {}: {}'.format(k, e, r, what)
  a.close(); b.close()


@pytest.mark.parametrize('n', [2048, 5000, 10000])
def test_ordered_launch_is_a_sorted_permutation_at_production_sizes(ref_pool, n):
  """The sizes the ordered launch switches itself on at (>= 2,048 envs: several pairs per thread of the sorting network;
  > 8,192: 128 KB of dynamic LDS): the permutation the settle kernel is served is read back — every env exactly once, keys
  ascending (highest release first, ties in index order) — and an env's key is the release height `Observer.pose` computes
  for its action (the max over the rock's pixels of H[window] + O, observer.py:405-413): checked against the maps."""
  from stackrl_amd import env as envs
  L = 3
  g = envs.VecStackEnv(n_parallel=n, seed=41, pool=ref_pool, block=True, episode_length=L)      # ordered by batch size
  g.reset()
  g.kernel_times(); g.set_profiling(True)
  for k in range(2):
    a = g.sample()
    Hm, Om, _ = g.maps()                     # the state the keys are computed from
    g.step(a)
    keys, order = g.launch_order()
    assert np.array_equal(np.sort(order), np.arange(n)), 'call {}: not a permutation'.format(k)
    assert np.array_equal(keys & np.uint64(0xffffffff), np.arange(n, dtype=np.uint64)), 'key i carries env i'
    ks = keys[order]
    assert (ks[1:] > ks[:-1]).all(), 'call {}: keys not strictly ascending along the order'.format(k)
    an = a.cpu().numpy()
    AW = 128 - 32 + 1
    for e in np.random.RandomState(k).choice(n, 64, replace=False):
      u, v = int(an[e]) // AW, int(an[e]) % AW
      o = Om[e]
      z = np.float32(np.max(np.where(o > 1e-4, Hm[e, u:u + 32, v:v + 32] + o, np.float32(0))))
      want = (~np.float32(z).view(np.uint32)) & np.uint32(0xffffffff)
      assert int(keys[e] >> np.uint64(32)) == int(want), 'env {}: key {} != release height {}'.format(e, keys[e] >> np.uint64(32), z)
  # the two order kernels are timed on their own, not inside the settle kernel's figure: two launches per ordered step
  ms, nl = g.kernel_times()
  oms, onl = g.order_kernel_times()
  assert int(nl[0]) == 2 and int(nl[1]) == 2 and int(nl[2]) == 0 and onl == 4 and 0.0 < oms < 10.0, (ms, nl, oms, onl)
  g.close()


@pytest.mark.parametrize('cap,smooth', [(3, True), (12, True), (9, False)])
def test_step_cap_exits_match_the_oracle(ref_pool, oracle_mod, cap, smooth):
  """The `MAX_STEP_TIME` cap (simulator.py:46, :221-224, :242-245; pinned for the oracle by tests/test_simulator_golden.py):
  with a cap of a few sub-steps some envs leave the smooth-placing loop at it, others the settle loop, others finish below
  it.  The reference raises RuntimeError; the kernel flags the env (SRL_ST_DIVERGED), `step` raises, and the state it leaves
  — counters, poses, place poses' effect on the next reward — equals the oracle's bit for bit."""
  from stackrl_amd import env as envs
  from stackrl_amd.config import StackConfig
  n, L = 24, 5
  kw = dict(max_substeps=cap, smooth_placing=smooth)
  g = envs.VecStackEnv(n_parallel=n, seed=3, pool=ref_pool, block=True, episode_length=L, **kw)
  o = oracle_mod.OracleEnv(StackConfig(n_envs=n, episode_length=L, **kw), ref_pool, seed=3)
  g.reset(); o.reset()
  seen = set()
  for k in range(L):
    ga, oa = g.sample(), o.sample()
    assert np.array_equal(ga.cpu().numpy(), oa)
    o.step(oa)
    try:
      g.step(ga)
      raised = False
    except RuntimeError:
      raised = True
    gp, gnb, gsub, gst = g.state()
    op, onb, osub, ost = o.state()
    assert np.array_equal(gst, ost) and np.array_equal(gsub, osub) and np.array_equal(gnb, onb), 'step {}'.format(k)
    assert raised == bool((ost & 1).any())
    np.testing.assert_allclose(gp, op, rtol=0, atol=POSE_ATOL)
    gH, _, _ = g.maps(); oH, _, _ = o.maps()
    assert np.array_equal(gH, oH)
    for e in range(n):
      if ost[e] & 1:
        seen.add('smooth' if osub[e, 1] == 0 and smooth else 'settle')
      else:
        seen.add('below')
  assert 'below' in seen or cap < 6
  assert ('smooth' in seen or 'settle' in seen), seen
  g.close()


def test_full_size_batch_properties(ref_pool):
  """BASELINE configs[1] size (1,024 envs x 8 rocks): size-independent properties instead of the oracle —
  every env places exactly L rocks, done on the L-th step, rocks at rest above the ground inside the time cap,
  height map consistent with the packed observation, rewards telescope to the final IoU."""
  from stackrl_amd import env as envs
  B, L = 1024, 8
  g = envs.VecStackEnv(n_parallel=B, seed=5, pool=ref_pool, block=True, episode_length=L)
  g.reset()
  total = torch.zeros(B, device='cuda')
  for k in range(L):
    (om, oo), r, d = g.step(g.sample())
    total += r
    assert bool(d.all()) == (k == L - 1) and bool(d.any()) == (k == L - 1)
  poses, nb, sub, st = g.state()
  assert (nb == L).all() and (st == 0).all() and (sub.sum(1) < 3000).all()
  v = g.velocities()
  assert np.linalg.norm(v[..., :3], axis=-1).max() <= 0.01 + 1e-7
  assert (poses[:, :, 2][:, :L] > 0.004).all()
  Hm, Om, goal = g.maps()
  assert np.array_equal(om[..., 0].cpu().numpy(), (Hm * np.float32(255) / np.float32(0.375)).astype(np.uint8))
  G = np.zeros_like(Hm)
  for i, (u, v_, h, w) in enumerate(goal):
    G[i, u:u + h, v_:v_ + w] = 0.25
  iou = np.minimum(Hm, 0.25)[G > 0].reshape(B, -1).sum(1) / np.maximum(Hm, G).reshape(B, -1).sum(1) if len(set((goal[:, 2] * goal[:, 3]).tolist())) == 1 else \
      np.array([np.minimum(Hm[i], 0.25)[G[i] > 0].sum() / np.maximum(Hm[i], G[i]).sum() for i in range(B)])
  np.testing.assert_allclose(total.cpu().numpy(), iou, rtol=2e-4, atol=2e-6)    # sum of reward differences = final metric
  assert float(oo.max()) == 0                                                     # terminal observation: nothing pending
  g.close()


@pytest.mark.parametrize('B,L,kw', [
  (4096, 16, {}),                                # BASELINE configs[2] per-GPU size
  (2048, 16, {}),                                # BASELINE configs[3] per-GPU shard (8,192 envs over 4 GPUs): four-wave variant + ordered launch
  (2048, 32, dict(resolution_factor=4)),         # BASELINE configs[4] per-GPU size: 32 rocks, 64 x 64 maps
])
def test_full_size_dqn_config_properties(B, L, kw):
  """The env step at the sizes of the DQN configs, through size-independent properties (the oracle would take hours):
  episode machine, rest criterion, bounded sub-steps, observation bytes = packed height map, telescoping rewards, and
  determinism — the first 64 envs of the big batch equal a 64-env batch with the same seed (envs are independent,
  utils.py:424-448, keys seed + i) bit for bit."""
  from stackrl_amd import assets, env as envs
  pool = assets.default_pool()
  g = envs.VecStackEnv(n_parallel=B, seed=5, pool=pool, block=True, episode_length=L, **kw)
  s = envs.VecStackEnv(n_parallel=64, seed=5, pool=pool, block=True, episode_length=L, **kw)
  (om, oo), _, _ = g.reset()
  (sm, so), _, _ = s.reset()
  assert torch.equal(om[:64], sm) and torch.equal(oo[:64], so)
  total = torch.zeros(B, device='cuda')
  for k in range(L):
    a = g.sample()
    assert torch.equal(a[:64], s.sample())
    (om, oo), r, d = g.step(a)
    (sm, so), sr, sd = s.step(a[:64])
    assert torch.equal(om[:64], sm) and torch.equal(oo[:64], so) and torch.equal(r[:64], sr) and torch.equal(d[:64], sd)
    total += r
    assert bool(d.all()) == (k == L - 1) and bool(d.any()) == (k == L - 1)
  poses, nb, sub, st = g.state()
  assert np.array_equal(poses[:64], s.state()[0])
  assert (nb == L).all() and (st == 0).all() and (sub.sum(1) < 6000).all()
  v = g.velocities()
  assert np.linalg.norm(v[..., :3], axis=-1).max() <= 0.01 + 1e-7
  assert (poses[:, :L, 2] > 0.004).all()
  Hm, Om, goal = g.maps()
  assert np.array_equal(om[..., 0].cpu().numpy(), (Hm * np.float32(255) / np.float32(0.375)).astype(np.uint8))
  iou = np.array([np.minimum(Hm[i], 0.25)[u:u + h, v_:v_ + w].sum() /
                  (np.maximum(Hm[i], 0).sum() + 0.25 * h * w - np.minimum(Hm[i], 0.25)[u:u + h, v_:v_ + w].sum())
                  for i, (u, v_, h, w) in enumerate(goal)])
  np.testing.assert_allclose(total.cpu().numpy(), iou, rtol=5e-4, atol=5e-6)    # sum of reward differences = final IoU
  assert float(oo.max()) == 0
  g.close(); s.close()


def test_input_constraints_are_reported(ref_pool):
  """`max_z` above 4 m has no codec table; a mesh that is not a closed two-manifold has no outline."""
  import copy
  from stackrl_amd import env as envs
  with pytest.raises(ValueError, match='max_z'):
    envs.VecStackEnv(n_parallel=2, seed=0, pool=ref_pool, episode_length=2, max_z=4.5)
  g = envs.VecStackEnv(n_parallel=2, seed=0, pool=ref_pool, episode_length=2, max_z=1.0)   # a taller window is fine
  g.reset(); g.step(g.sample()); g.close()
  bad = copy.copy(ref_pool)
  bad.tris = ref_pool.tris.copy()
  bad.tris[int(ref_pool.tri_off[0])] = bad.tris[int(ref_pool.tri_off[0]) + 1]      # a face listed twice: an edge with 3 faces
  with pytest.raises(ValueError):
    envs.VecStackEnv(n_parallel=2, seed=0, pool=bad, episode_length=2)


def _banded(shape, dtype, band=4096):
  """A caller buffer with canary bands on both sides: (view to hand to the env, the whole allocation, pattern)."""
  n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
  pat = 0xA5
  raw = torch.full((band + n + band,), pat, dtype=torch.uint8, device='cuda')
  return raw[band:band + n].view(dtype).view(shape), raw, band, n


def _bands_intact(raw, band, n):
  return bool((raw[:band] == 0xA5).all()) and bool((raw[band + n:] == 0xA5).all())


def test_two_handles_on_two_streams_equal_one_handle_and_the_oracle(ref_pool, oracle_mod):
  """include/stackrl_hip.h: "distinct handles are independent" (the reference's workers are separate processes,
  utils.py:424-448).  Two handles of 24 envs, each on its own stream, step CONCURRENTLY (both launched before either is
  waited for) over RNG-driven episodes and their auto-reset, next to a one-handle env of the same 48 envs and the oracle:
  observations, rewards, done flags bit-identical three ways, states equal; the caller buffers of the two handles are
  slices of one allocation with canary bands around it, which stay untouched (no env kernel writes outside what it was
  given).  Round 2 had one failing record of a four-handle run against the one-handle env (gpurun_out/t_learner.log); the
  handles then shared process-wide `__constant__` pair tables rewritten by every srl_create and a thread-local reset
  scratch that was freed and re-allocated on growth — both gone."""
  from stackrl_amd import env as envs
  from stackrl_amd.config import StackConfig
  B, G, L, seed = 48, 24, 6, 77
  a = envs.VecStackEnv(n_parallel=B, seed=seed, pool=ref_pool, episode_length=L)
  o = oracle_mod.OracleEnv(StackConfig(n_envs=B, episode_length=L), ref_pool, seed=seed)
  shards = [envs.VecStackEnv(n_parallel=G, seed=seed, pool=ref_pool, episode_length=L, env_index_offset=k * G,
                             side_stream=True) for k in range(2)]
  assert a.seed(seed) == shards[0].seed(seed) + shards[1].seed(seed)
  spec = a.observation_spec

  def buffers():
    return [_banded((B,) + tuple(spec[0].shape), torch.uint8), _banded((B,) + tuple(spec[1].shape), torch.uint8),
            _banded((B,), torch.float32), _banded((B,), torch.uint8)]

  bufs = buffers()
  om, oo = bufs[0][0], bufs[1][0]
  waits = [s.reset(block=False, out=(om[k * G:(k + 1) * G], oo[k * G:(k + 1) * G])) for k, s in enumerate(shards)]
  sa = a.reset()()
  so = o.reset()
  for w in waits:
    w()
  assert torch.equal(sa[0][0], om) and torch.equal(sa[0][1], oo)
  assert np.array_equal(om.cpu().numpy(), so[0][0]) and np.array_equal(oo.cpu().numpy(), so[0][1])
  assert all(_bands_intact(*b[1:]) for b in bufs[:2])
  filler = torch.randn(1024, 1024, device='cuda')
  for t in range(2 * (L + 1)):
    act = a.sample()
    assert np.array_equal(act.cpu().numpy(), o.sample())
    assert torch.equal(act, torch.cat([s.sample() for s in shards]))
    bufs = buffers()
    om, oo, r, d = (b[0] for b in bufs)
    waits = []
    for k, s in enumerate(shards):        # both shards in flight, with current-stream work in between, before anything is waited for
      waits.append(s.step(act[k * G:(k + 1) * G], block=False,
                          out=(om[k * G:(k + 1) * G], oo[k * G:(k + 1) * G], r[k * G:(k + 1) * G], d[k * G:(k + 1) * G])))
      filler = (filler @ filler) * 1e-3
    sa = a.step(act)()
    for w in waits:
      w()
    tag = 'call {}'.format(t)
    for nm, x, y in (('obs_map', sa[0][0], om), ('obs_obj', sa[0][1], oo), ('reward', sa[1], r), ('done', sa[2], d.view(torch.bool))):
      if not torch.equal(x, y):
        bad = (x != y).reshape(B, -1).any(1).nonzero()[:, 0].tolist()
        raise AssertionError('{}: {} of the two-handle run differs from the one-handle env in envs {}'.format(tag, nm, bad[:16]))
    assert all(_bands_intact(*b[1:]) for b in bufs), tag + ': an env kernel wrote outside its caller buffer'
    _cmp_step(a, o, sa, o.step(act.cpu().numpy()), tag)
    ps = [s.state() for s in shards]
    pa = a.state()
    for i in range(4):
      assert np.array_equal(pa[i], np.concatenate([p[i] for p in ps])), tag + ': state {}'.format(i)
  a.close()
  for s in shards:
    s.close()


def test_pipelined_env_equals_one_handle_and_the_oracle(ref_pool, oracle_mod):
  """`PipelinedVecStackEnv` (the batch as groups of handles that step as their actions arrive): through its plain `step` and
  through `collect_step` the 12 envs x 12 rocks follow the one-handle env and the oracle bit for bit over two episodes and
  their auto-reset; the groups run the throughput-oriented settle build (`concurrent_envs` says 4,096 envs share the
  device) while the one-handle env of 12 runs the latency-oriented one — results do not depend on the build."""
  from stackrl_amd import env as envs
  from stackrl_amd.config import StackConfig
  B, L, seed = 12, 12, 31
  a = envs.make('Stack-v0', n_parallel=B, seed=seed, pool=ref_pool, episode_length=L)
  p = envs.make('Stack-v0', n_parallel=B, seed=seed, pool=ref_pool, episode_length=L, groups=(2, 4, 6), concurrent_envs=4096)   # unequal groups
  q = envs.make('Stack-v0', n_parallel=B, seed=seed, pool=ref_pool, episode_length=L, groups=2)
  o = oracle_mod.OracleEnv(StackConfig(n_envs=B, episode_length=L), ref_pool, seed=seed)
  assert p.groups == 3 and p.batch_size == B and p.n_actions == a.n_actions
  assert p._envs[0]._lib is not None and a.seed(seed) == p.seed(seed) == q.seed(seed)
  sa, sp = a.reset()(), p.reset()()
  q.reset()                                    # not waited for: collect_step takes the waits group by group
  so = o.reset()
  assert torch.equal(sa[0][0], sp[0][0]) and torch.equal(sa[0][1], sp[0][1])
  assert np.array_equal(sp[0][0].cpu().numpy(), so[0][0])
  prev = sa
  for t in range(2 * (L + 1)):
    act = a.sample()
    assert torch.equal(act, p.sample()) and torch.equal(act, q.sample())
    assert np.array_equal(act.cpu().numpy(), o.sample())
    seen = []
    step_q, act_q = q.collect_step(lambda k, s, st: (seen.append((k, s, st)), act[s])[1])
    assert torch.equal(act_q, act) and [k for k, _, _ in seen] == [0, 1]
    tag = 'call {}'.format(t)
    for x, y in zip((prev[0][0], prev[0][1], prev[1], prev[2]), (step_q[0][0], step_q[0][1], step_q[1], step_q[2])):
      assert torch.equal(x, y), tag + ': the step collect_step hands the policy is the one-handle env\'s latest step'
    for k, s, st in seen:                      # and every group saw its own slice of it
      assert torch.equal(st[0][0], prev[0][0][s]) and torch.equal(st[1], prev[1][s]) and torch.equal(st[2], prev[2][s])
    sa, sp = a.step(act)(), p.step(act)()
    for nm, x, y in (('obs_map', sa[0][0], sp[0][0]), ('obs_obj', sa[0][1], sp[0][1]), ('reward', sa[1], sp[1]), ('done', sa[2], sp[2])):
      assert torch.equal(x, y), '{}: {} of the pipelined env differs from the one-handle env'.format(tag, nm)
    _cmp_step(a, o, sa, o.step(act.cpu().numpy()), tag)
    for i, (u, v) in enumerate(zip(a.state(), p.state())):
      assert np.array_equal(u, v), tag + ': state {}'.format(i)
    prev = sa
  q.drain()
  for i, (u, v) in enumerate(zip(a.state(), q.state())):
    assert np.array_equal(u, v), 'final state {} after collect_step'.format(i)
  for e in (a, p, q):
    e.close()


@pytest.mark.parametrize('B,L,episodes,hint', [(512, 3, 2, None), (256, 8, 1, None), (96, 16, 1, 4096)])
def test_env_step_under_the_concurrent_rollout_forward_equals_the_oracle(oracle_mod, B, L, episodes, hint):
  """The training loop steps the envs on a side stream while the Q-net's kernels run on the current one; the env's results
  must not depend on that.  512 envs x 3 rocks over two episodes (and 8 rocks; 16 rocks on the throughput-oriented settle
  build, `concurrent_envs`), every step in flight while the rollout forward (the hand-written convolutions, the
  cross-correlation, the policy head) runs beside it, against the oracle in lock step: bytes, rewards, poses, sub-step
  counts and height maps bit for bit.
  This is the reproduction of round 2's "concurrency anomalies" (DESIGN.md section 6a): built with clang's SLP vectoriser
  (packed-fp32 code in the settle kernel) the env returned results that differ from the oracle's in 20 - 50 env steps of
  these 4,096 — only under this concurrency, never alone; csrc is built with -fno-slp-vectorize since (stackrl_amd/build.py)."""
  from stackrl_amd import assets, env as envs, nets, qops
  from stackrl_amd.config import StackConfig
  seed = 5
  pool = assets.default_pool()
  e = envs.make('Stack-v0', n_parallel=B, seed=seed, pool=pool, episode_length=L, side_stream=True,
                **({} if hint is None else dict(concurrent_envs=hint)))
  o = oracle_mod.OracleEnv(StackConfig(n_envs=B, episode_length=L), pool, seed=seed)
  net = nets.DeepQSiamFCN(e.observation_spec, seed=2).cuda()
  pol = qops.FusedPolicy(chunk=256, fast=True)
  gen = torch.Generator(device='cuda').manual_seed(1)
  step = e.reset()()
  o.reset()
  bad = []
  for t in range(episodes * (L + 1)):
    a = e.sample()
    ao = o.sample()
    assert np.array_equal(a.cpu().numpy(), ao)
    w = e.step(a, block=False)
    for _ in range(max(1, 512 // B)):
      pol(net, step[0], 1.0, gen)             # the forward on the previous observation, while the step is in flight
    step = w()
    (om, oo), r, d = step
    (omo, ooo), ro, do = o.step(ao)
    gs, os_ = e.state(), o.state()
    gH, oH = e.maps()[0], o.maps()[0]
    same = (np.array_equal(om.cpu().numpy(), omo) and np.array_equal(oo.cpu().numpy(), ooo)
            and np.array_equal(r.cpu().numpy().view(np.uint32), ro.view(np.uint32)) and np.array_equal(d.cpu().numpy(), do)
            and all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
                    for x, y in zip(gs, os_))
            and np.array_equal(gH.view(np.uint32), oH.view(np.uint32)))
    if not same:
      envs_bad = np.nonzero((gs[0].view(np.uint32) != os_[0].view(np.uint32)).reshape(B, -1).any(1))[0]
      bad.append((t, envs_bad[:8].tolist()))
  e.close()
  assert not bad, 'env results under the concurrent forward differ from the oracle (call, envs): {}'.format(bad)
