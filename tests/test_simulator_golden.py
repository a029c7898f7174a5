"""`Simulator.step`'s control flow (P1 - P4) pinned by the reference's own simulator.py.

tests/golden/make_simulator_golden.py ran /root/reference/stackrl/envs/stack/simulator.py AS IT IS over a recording pybullet
placeholder whose physics is a script (speeds, contact counts and poses per stepSimulation call: inputs of the fixture) and
recorded every call the reference made of it, `n_steps`, the RuntimeError exits, the place / final poses and
`distances_from_place`.  Here the oracle's `sim_step_world` — the function `srlo_step` runs its own rigid-body step under
(oracle/srl_oracle.c) — is driven by the same scripts: the sequence of calls must be the reference's, call for call.

What this does NOT pin: what pybullet computes between the calls (the rigid-body step, the contact points, the difference
quaternion) — those rows stay "parity unpinned" (DESIGN.md section 2).
"""
import ctypes
import os

import numpy as np
import pytest

from oracle import oracle as orc
from stackrl_amd.config import StackConfig

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'simulator_golden.npz'))
CASES = [(c, s) for c in range(int(G['n_cases'])) for s in range(int(G['c{}_n_calls'.format(c)]))]


def _lib():
  L = orc.lib()
  I32, F, VP = ctypes.c_int32, ctypes.c_float, ctypes.c_void_p
  L.srlo_sim_step_scripted.restype = ctypes.c_int
  L.srlo_sim_step_scripted.argtypes = [I32, I32, I32, F, I32, I32, I32, VP, VP, VP, VP, VP, I32]
  L.srlo_distance_from_place.restype = None
  L.srlo_distance_from_place.argtypes = [VP, VP, VP]
  L.srlo_max_substeps.restype = I32
  L.srlo_max_substeps.argtypes = [VP]
  return L


def _p(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def _reference_calls(log, n_before, has_new):
  """The reference's log in the oracle's vocabulary.  Of its getBasePositionAndOrientation calls (5, b) only the first of a
  step is a call of the WORLD's interface (the place pose of the newest body, simulator.py:227); the others read the poses of
  all bodies before and after the settle loop (simulator.py:234-237, :250-253) — state the oracle holds itself — and the
  trailing loadURDF (9) belongs to `_load`."""
  out, i, first = [], 0, True
  newest = n_before + (1 if has_new else 0) - 1
  poses_asked = []
  while i < len(log):
    c = int(log[i])
    if c == 5:
      b = int(log[i + 1])
      if first:
        assert b == newest, 'the place pose is that of the newest body'
        out.append(5)
        first = False
      else:
        poses_asked.append(b)
      i += 2
      continue
    if c != 9:
      out.append(c)
    i += 1
  return out, poses_asked


@pytest.mark.parametrize('c,s', CASES)
def test_sim_step_makes_the_reference_calls(c, s):
  L = _lib()
  p = 'c{}_s{}_'.format(c, s)
  speeds = np.ascontiguousarray(G[p + 'speeds'], np.float32)
  contacts = np.ascontiguousarray(G[p + 'contacts'], np.int32)
  K, nb = speeds.shape
  has_new, n_before, smooth, raised = (int(G[p + k]) for k in ('has_new', 'n_before', 'smooth', 'raised'))
  cap = int(G['c{}_max_step_count'.format(c)])
  log = np.zeros(4096, np.int32)
  sub = np.zeros(2, np.int32)
  r = ctypes.c_int32(-1)
  n = L.srlo_sim_step_scripted(has_new, n_before, smooth, float(G['c{}_velocity_threshold'.format(c)]), cap, K, nb, _p(speeds),
                               _p(contacts), _p(sub), ctypes.byref(r), _p(log), len(log))
  assert n > 0
  want, poses_asked = _reference_calls(G[p + 'log'], n_before, has_new)
  assert r.value == raised
  if raised:
    # the reference raises out of the loop: nothing after the failing stepSimulation
    assert log[:n].tolist() == want
    return
  assert log[:n].tolist() == want, 'calls differ: oracle {} reference {}'.format(log[:n].tolist(), want)
  assert sub.tolist() == G[p + 'n_steps'].tolist()
  assert int(sub.sum()) == int(G[p + 'steps_taken']) + (0 if has_new else 1)      # the counter starts at 1 with or without _place
  # the reference then reads every body's pose twice (initial, final), oldest first
  assert poses_asked == list(range(nb)) * 2


def test_step_counter_semantics_of_the_fixture():
  """The fixture holds each exit of each loop (make_simulator_golden.py cases A - G)."""
  kinds = set()
  for c, s in CASES:
    p = 'c{}_s{}_'.format(c, s)
    log = G[p + 'log'].tolist()
    if int(G[p + 'raised']):
      kinds.add('cap in the smooth loop' if 3 in log and log[-2:] != [2] and log[-2] == 3 else 'cap in the settle loop')
      continue
    a, b = G[p + 'n_steps'].tolist()
    smooth = int(G[p + 'smooth'])
    kinds.add(('smooth' if smooth else 'plain') + (' S_a>1' if a > 1 else ' S_a=1') + (' S_b>0' if b > 0 else ' S_b=0'))
    if not int(G[p + 'has_new']):
      kinds.add('nothing waiting')
  assert {'cap in the smooth loop', 'cap in the settle loop', 'smooth S_a>1 S_b>0', 'smooth S_a=1 S_b=0', 'smooth S_a>1 S_b=0',
          'plain S_a=1 S_b>0', 'plain S_a=1 S_b=0', 'nothing waiting'} <= kinds, kinds


@pytest.mark.parametrize('c,s', [cs for cs in CASES if not int(G['c{}_s{}_raised'.format(*cs)])])
def test_distances_from_place(c, s):
  """(|p_place - p_now|, 2 acos(min(w, 1))) of every rock: the reference's numbers from the fixture's poses against the
  oracle's `distance_from_place` on the same poses (float32; the oracle takes w = |q_place . q_now|, which is what the
  placeholder's difference quaternion hands the reference: recorded as `w_handed_out`)."""
  L = _lib()
  p = 'c{}_s{}_'.format(c, s)
  place, final, want = G[p + 'place_poses'], G[p + 'final_poses'], G[p + 'distances_from_place']
  w = G[p + 'w_handed_out']
  # with nothing waiting the reference still appends a place pose (simulator.py:227) and `zip` drops the surplus entry
  assert len(final) == len(want) == len(w) and len(place) == len(final) + (0 if int(G[p + 'has_new']) else 1)
  for b in range(len(final)):
    out = np.zeros(2, np.float32)
    L.srlo_distance_from_place(_p(np.ascontiguousarray(place[b], np.float32)), _p(np.ascontiguousarray(final[b], np.float32)), _p(out))
    assert abs(float(out[0]) - want[b, 0]) <= 2e-7 + 1e-6 * want[b, 0]
    # the reference's own formula on the w it was handed
    assert abs(2 * np.arccos(min(w[b], 1.0)) - want[b, 1]) <= 1e-12
    assert abs(abs(float(np.dot(place[b, 3:], final[b, 3:]))) - w[b]) <= 1e-12
    # float32 quaternions: acos is ill-conditioned near w = 1 (d angle = dw / sin(angle / 2)); bound by the float32 rounding of w
    tol = 4e-7 / max(np.sqrt(max(1 - w[b] ** 2, 0.0)), 1e-3) + 1e-6
    assert abs(float(out[1]) - want[b, 1]) <= tol, (float(out[1]), want[b, 1], tol)


def test_step_cap_is_the_reference_value():
  """`int(MAX_STEP_TIME / time_step)` (simulator.py:6, :46) for the time steps in use: what the host mirror hands the library
  (computed in Python's doubles, like the reference) and what the C side derives from the float32 time step alone."""
  L = _lib()
  for ts, cap in zip(G['cap_time_steps'].tolist(), G['cap_values'].tolist()):
    cfg = StackConfig(sim_time_step=ts)
    c = cfg.to_c()
    assert c.max_substeps == cap, (ts, c.max_substeps, cap)
    c.max_substeps = 0
    assert L.srlo_max_substeps(ctypes.byref(c)) == cap, ts


def test_reset_sets_the_world_up_like_the_reference():
  """`Simulator.reset` (simulator.py:156-188): connect + time step on first use, gravity, the ground box; recorded by name."""
  assert G['c0_setup'].tolist() == ['connect', 'setTimeStep', 'setGravity', 'createCollisionShape', 'createVisualShape', 'createMultiBody']
