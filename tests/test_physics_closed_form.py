"""Closed-form pins of the settle solver (rows P1-P5 of SURVEY.md section 8a).

pybullet is not pinned by the reference, so analytic mechanics is what the rigid-body rows can
be held to: free fall, the rest height of one and of two stacked cuboids, the Coulomb sliding distance on the ground
and the smooth-placing step count of `Simulator.step` (simulator.py:212-224).  The expectations are committed in
tests/golden/physics_closed_form.json, written by tests/golden/make_physics_closed_form.py in float64 without any build
code; the same scenarios run on the CPU oracle (here, `-m "not gpu"`) and on the HIP kernels through the C-ABI
(`-m gpu`), which must also agree with the oracle bit for bit.
"""
import importlib.util
import json
import os

import numpy as np
import pytest

from stackrl_amd.config import StackConfig

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'physics_closed_form.json')) as _f:
  CF = json.load(_f)
_spec = importlib.util.spec_from_file_location('make_physics_closed_form',
                                               os.path.join(HERE, 'golden', 'make_physics_closed_form.py'))
_gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gen)

CUBOIDS = [64, 65, 66]            # `0_0`, `0_1`, `0_2` of tests/golden/ref_rocks.npz
CENTRE = 48 * 97 + 48             # action: object window centred on the 128 x 128 map (env.py:240-241)
HZ = CF['params']['cuboid_half_extents'][2]


class _Sim(object):
  """One env behind the calls both sides share; `kind` = 'oracle' or 'hip'."""

  def __init__(self, kind, oracle_mod, pool, L=3):
    self.kind = kind
    if kind == 'oracle':
      self.env = oracle_mod.OracleEnv(StackConfig(n_envs=1, episode_length=L), pool, seed=1)
    else:
      from stackrl_amd import env as envs
      self.env = envs.VecStackEnv(n_parallel=1, seed=1, pool=pool, block=True, episode_length=L)
    self.env.set_script(np.array([CUBOIDS[:L]], np.int32), np.array([[40, 40, 64, 64]], np.int32))
    self.env.reset()

  def place(self, action=CENTRE):
    a = np.array([action], np.int64)
    if self.kind == 'hip':
      import torch
      a = torch.from_numpy(a).cuda()
    self.env.step(a)

  def __getattr__(self, name):
    return getattr(self.env, name)


def _rest_heights(sim):
  sim.place()
  assert sim.state()[2][0, 0] == CF['s_a']['0.0000']          # released in touch with the ground: S_a = 1
  sim.step_simulation(300)
  z1 = float(sim.state()[0][0, 0, 2])
  speed = np.linalg.norm(sim.velocities()[0, 0, :3])
  pen1 = float(sim.contacts()[0][0])
  sim.place()
  sim.step_simulation(300)
  p = sim.state()[0][0]
  return z1, speed, pen1, float(p[0, 2]), float(p[1, 2]), float(sim.contacts()[0][0])


def _check_rest(sim):
  z1, speed, pen1, z1b, z2, pen2 = _rest_heights(sim)
  tol = CF['rest_tol']
  assert abs(z1 - CF['rest_z_one']) <= tol, (z1, CF['rest_z_one'])
  assert speed < 1e-3
  assert pen1 <= CF['params']['linear_slop'] + 2e-6              # at rest the only penetration left is the slop
  assert abs(z1b - CF['rest_z_one']) <= tol
  assert abs(z2 - CF['rest_z_two']) <= tol, (z2, CF['rest_z_two'])
  assert pen2 <= CF['params']['linear_slop'] + 2e-5              # the loaded lower contact sinks a few micrometres more


def _check_free_fall(sim):
  sim.place()
  p = sim.state()[0].copy()
  ff = CF['free_fall']
  p[0, 0, :7] = [0.25, 0.25, ff['z0'], 0, 0, 0, 1]
  sim.set_body_state(p, np.zeros_like(p))
  zs = []
  for _ in range(len(ff['z'])):
    sim.step_simulation(1)
    zs.append(float(sim.state()[0][0, 0, 2]))
  assert np.abs(np.array(zs) - np.array(ff['z'])).max() <= ff['tol']
  q = sim.state()[0][0, 0, 3:7]
  assert np.array_equal(q, np.array([0, 0, 0, 1], np.float32))   # no torque on a free body


def _check_slide(sim):
  sim.place()
  sim.step_simulation(300)
  p = sim.state()[0].copy()
  z = p[0, 0, 2]
  sl = CF['slide']
  p[0, 0, :7] = [0.1, 0.25, z, 0, 0, 0, 1]
  v = np.zeros_like(p)
  v[0, 0, 0] = sl['v0']              # along +x: a friction axis of the ground contact (plane_space of n = +z)
  sim.set_body_state(p, v)
  sim.step_simulation(sl['substeps'] + 100)
  x = sim.state()[0][0, 0]
  dist = float(x[0]) - 0.1
  assert abs(dist - sl['distance']) <= sl['tol_rel'] * sl['distance'], (dist, sl['distance'])
  assert abs(dist - sl['continuum']) <= 0.05 * sl['continuum']   # mu = 0.3 sliding distance v0^2 / (2 mu g) +- 5 %
  assert abs(float(x[1]) - 0.25) < 1e-3 and abs(float(x[2]) - z) < 1e-4
  assert np.linalg.norm(sim.velocities()[0, 0, :3]) < 1e-3


def _check_smooth_placing(sim, stack):
  """Release a flat cuboid h above the bare ground: the plateau in the stale height map belongs to `stack` cuboids that
  were teleported away after the map was rendered (`Observer.pose` reads the map of the previous step)."""
  for _ in range(stack):
    sim.place()
    sim.step_simulation(200)
  # the map was rendered by the last place(), before the extra sub-steps (which move the stack by micrometres only)
  Hm, Om, _ = sim.maps()
  Hm, Om = Hm[0], Om[0].reshape(32, 32)
  win = Hm[48:80, 48:80]
  z_place = float((win + Om)[Om > 1e-4].max()) - 0.0625          # observer.py:405-413
  h = z_place - HZ
  p = sim.state()[0].copy()
  for b in range(stack):
    p[0, b, :3] = [0.44, 0.06, p[0, b, 2]]                  # out of the way, same heights
  sim.set_body_state(p, np.zeros_like(p))
  sim.place()
  s_a = int(sim.state()[2][0, 0])
  expect = _gen.smooth_placing_steps(h)
  # the closed form must not sit on a rounding boundary for the comparison to be meaningful
  gdt2 = CF['params']['gravity'] * CF['params']['dt'] ** 2
  frac = ((h - CF['params']['margin'] - CF['params']['breaking_threshold']) / gdt2) % 1.0
  assert 0.02 < frac < 0.98, 'scenario sits on a boundary: {}'.format(frac)
  assert s_a == expect, (s_a, expect, h)
  return s_a


def test_generator_reproduces_committed_json():
  assert _gen.free_fall(0.3, 20) == CF['free_fall']['z']
  assert _gen.slide(CF['slide']['v0'])[0] == CF['slide']['distance']
  assert {k: _gen.smooth_placing_steps(float(k)) for k in CF['s_a']} == CF['s_a']


def test_oracle_rest_heights(oracle_mod, ref_pool):
  _check_rest(_Sim('oracle', oracle_mod, ref_pool))


def test_oracle_free_fall(oracle_mod, ref_pool):
  _check_free_fall(_Sim('oracle', oracle_mod, ref_pool))


def test_oracle_sliding_distance(oracle_mod, ref_pool):
  _check_slide(_Sim('oracle', oracle_mod, ref_pool))


@pytest.mark.parametrize('stack', [1, 2])
def test_oracle_smooth_placing_count(oracle_mod, ref_pool, stack):
  s_a = _check_smooth_placing(_Sim('oracle', oracle_mod, ref_pool), stack)
  assert s_a > 30 * stack


@pytest.mark.gpu
def test_hip_closed_forms_and_oracle_bits(oracle_mod, ref_pool):
  pytest.importorskip('torch')
  _check_rest(_Sim('hip', oracle_mod, ref_pool))
  _check_free_fall(_Sim('hip', oracle_mod, ref_pool))
  _check_slide(_Sim('hip', oracle_mod, ref_pool))
  for stack in (1, 2):
    _check_smooth_placing(_Sim('hip', oracle_mod, ref_pool), stack)
  # and bit for bit against the oracle through the same injected states
  g, o = _Sim('hip', oracle_mod, ref_pool), _Sim('oracle', oracle_mod, ref_pool)
  for s in (g, o):
    s.place()
    s.step_simulation(7)
    p = s.state()[0].copy()
    p[0, 0, :7] = [0.2, 0.3, p[0, 0, 2] + 0.004, 0, 0, 0.38268343, 0.92387953]
    v = np.zeros_like(p)
    v[0, 0, :3] = [0.3, -0.2, 0.1]
    v[0, 0, 4:7] = [1.0, -2.0, 0.5]
    s.set_body_state(p, v)
    s.step_simulation(40)
    s.place()
  assert np.array_equal(g.state()[0], o.state()[0])
  assert np.array_equal(g.velocities(), o.velocities())
  assert np.array_equal(g.state()[2], o.state()[2])
  assert np.array_equal(g.sweeps(), o.sweeps())
