#!/usr/bin/env python3
"""Golden vectors for the CONTROL FLOW of the physics driver (P1 - P4) from the reference's OWN code:
`stackrl/envs/stack/simulator.py` is loaded by file path and run as it is.

Runs only in the build container (needs /root/reference).  The file imports `functools`, `numpy` and `pybullet`; pybullet
is what is absent, so `sys.modules['pybullet']` is a recording placeholder whose "physics" is a SCRIPT: after k calls of
`stepSimulation` inside one `Simulator.step`, body b moves at `speeds[k][b]`, the newest body has `contacts[k]` contact
points and body b sits at `poses[k][b]` — all of them INPUTS of the fixture (seeded / hand-made), none of them computed.
What the reference's code DOES with such a world is what gets pinned:

  * `_place`: one `resetBasePositionAndOrientation` + one `stepSimulation`, only when an object is waiting (simulator.py:310-320)
  * the smooth-placing loop: `_drop` = three contact points OR the stop criterion, `resetBaseVelocity(newest, 0, 0)` +
    `stepSimulation` per turn, the step counter and its cap -> RuntimeError (simulator.py:212-224, :337-341)
  * the place pose taken AFTER that loop (simulator.py:227)
  * the settle loop: `_stop` asks the bodies newest -> oldest for their LINEAR velocity only and returns at the first fast
    one, counter and cap -> RuntimeError (simulator.py:239-245, :322-335)
  * `n_steps` = (steps before the drop, steps after) (simulator.py:79-83, :230, :247)
  * `distances_from_place` = (|p_place - p_now|, 2 acos(min(w, 1))) with w the last component of what
    `getDifferenceQuaternion` returns (simulator.py:113-128) — the difference quaternion itself is pybullet's: the
    placeholder returns end * start^-1 of the nearer of q and -q and the fixture records the w it handed out.

Every call the reference makes of the placeholder is logged in order as an integer code:
  1 resetBasePositionAndOrientation   2 stepSimulation   3 resetBaseVelocity(newest, zeros)   4 getContactPoints(newest)
  5 getBasePositionAndOrientation(b) -> 5, b      16 + b getBaseVelocity(objects[b])      9 loadURDF
(set-up calls of `reset` — resetSimulation / setGravity / createMultiBody ... — are recorded by name, not in the step log).

The file written (`simulator_golden.npz`) holds data only: per case the constructor arguments and per `step` call the
script, the log, `n_steps`, whether RuntimeError was raised, the place / final poses and `distances_from_place`.
tests/test_simulator_golden.py runs the oracle's `sim_step_world` (oracle/srl_oracle.c, the function its env steps with)
over the same scripts.
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True   # the reference tree is read-only: no __pycache__ beside its files
import types

import numpy as np

REF = '/root/reference/stackrl/envs/stack/simulator.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'simulator_golden.npz')


class World(object):
  """The scripted world behind the placeholder module."""

  def __init__(self):
    self.connected = False
    self.bodies = []          # ids handed out by loadURDF, in order
    self.setup = []           # names of the set-up calls
    self.script = None        # dict(speeds [K, nb], contacts [K], poses [K, nb, 7])
    self.k = 0
    self.log = []
    self.w_out = []
    self.next_id = 1

  def row(self, a):
    return a[min(self.k, len(a) - 1)]


W = World()


def _idx(body):
  return W.order.index(body)


def make_pybullet():
  pb = types.ModuleType('pybullet')
  pb.GUI, pb.DIRECT, pb.GEOM_BOX, pb.COV_ENABLE_GUI = 1, 2, 3, 4

  def connect(mode):
    W.connected = True
    W.setup.append('connect')
    return 0

  def disconnect(physicsClientId=0):
    W.connected = False

  def isConnected(physicsClientId=0):
    return W.connected

  def named(name, ret=0):
    def f(*a, **kw):
      assert 'physicsClientId' in kw          # simulator.py:57-61: every call goes through the partial
      W.setup.append(name)
      return ret
    return f

  for n in ('setTimeStep', 'resetSimulation', 'setGravity', 'createMultiBody', 'createCollisionShape', 'createVisualShape',
            'configureDebugVisualizer'):
    setattr(pb, n, named(n))

  def loadURDF(urdf, pos, orn, physicsClientId=0):
    W.log.append(9)
    W.loaded.append((str(urdf), tuple(pos), tuple(orn)))
    b = W.next_id
    W.next_id += 1
    return b

  def resetBasePositionAndOrientation(body, pos, orn, physicsClientId=0):
    W.log.append(1)
    W.placed.append((tuple(float(v) for v in pos), tuple(float(v) for v in orn)))
    W.order.append(body)                      # the reference appends it to _objects next (simulator.py:318)

  def stepSimulation(physicsClientId=0):
    W.k += 1
    W.log.append(2)

  def resetBaseVelocity(body, lin, ang, physicsClientId=0):
    assert body == W.order[-1] and list(lin) == [0, 0, 0] and list(ang) == [0, 0, 0]
    W.log.append(3)

  def getContactPoints(body, physicsClientId=0):
    assert body == W.order[-1]
    W.log.append(4)
    return [None] * int(W.row(W.script['contacts']))

  def getBaseVelocity(body, physicsClientId=0):
    b = _idx(body)
    W.log.append(16 + b)
    s = float(W.row(W.script['speeds'])[b])
    d = W.script['dirs'][b]
    # the angular velocity is large on purpose: the stop criterion must ignore it (simulator.py:332-333)
    return tuple(s * d), (50.0, -60.0, 70.0)

  def getBasePositionAndOrientation(body, physicsClientId=0):
    b = _idx(body)
    W.log.extend([5, b])
    p = W.row(W.script['poses'])[b]
    return tuple(float(v) for v in p[:3]), tuple(float(v) for v in p[3:])

  def getDifferenceQuaternion(q0, q1, physicsClientId=0):
    a, b = np.asarray(q0, np.float64), np.asarray(q1, np.float64)
    if np.dot(a, b) < 0:
      b = -b
    # end * conj(start), (x, y, z, w)
    av, aw, bv, bw = -a[:3], a[3], b[:3], b[3]
    v = bw * av + aw * bv + np.cross(bv, av)
    w = bw * aw - np.dot(bv, av)
    W.w_out.append(float(w))
    return (float(v[0]), float(v[1]), float(v[2]), float(w))

  for f in (connect, disconnect, isConnected, loadURDF, resetBasePositionAndOrientation, stepSimulation, resetBaseVelocity,
            getContactPoints, getBaseVelocity, getBasePositionAndOrientation, getDifferenceQuaternion):
    setattr(pb, f.__name__, f)
  return pb


def load_reference():
  sys.modules['pybullet'] = make_pybullet()
  spec = importlib.util.spec_from_file_location('ref_simulator', REF)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def unit(v):
  v = np.asarray(v, np.float64)
  return v / np.linalg.norm(v)


def make_script(rng, nb, K, end_smooth, end_settle, slow_old=True, contacts_at_end=3, fast_old_until=None):
  """speeds / contacts / poses over K rows for nb bodies (the newest is the last).  The newest body has fewer than three
  contacts and is fast until row `end_smooth`; every body is slow from row `end_settle` on; `fast_old_until`: an OLDER body
  stays fast until that row while the newest is already slow (the criterion must ask all of them)."""
  thr = 0.01
  speeds = np.full((K, nb), 0.5 * thr, np.float64)
  contacts = np.zeros(K, np.int32)
  for k in range(K):
    if k < end_settle:
      speeds[k, nb - 1] = thr * (1.5 + rng.uniform(0, 3))
    if fast_old_until is not None and k < fast_old_until and nb > 1:
      speeds[k, 0] = thr * 2.0
      if k >= end_settle:
        speeds[k, nb - 1] = 0.2 * thr
    contacts[k] = contacts_at_end if k >= end_smooth else int(rng.randint(0, 3))
  poses = np.zeros((K, nb, 7), np.float64)
  base = rng.uniform(0.05, 0.45, size=(nb, 3))
  q0 = np.stack([unit(rng.normal(size=4)) for _ in range(nb)])
  for k in range(K):
    for b in range(nb):
      poses[k, b, :3] = base[b] + 0.002 * k * np.array([0.3, -0.2, -1.0]) * (b + 1) / nb
      q = unit(q0[b] + 0.01 * k * np.array([0.2, -0.1, 0.3, 0.05]))
      poses[k, b, 3:] = q if (k + b) % 3 else -q        # the sign flips: q and -q are the same rotation
  dirs = np.stack([unit(rng.normal(size=3)) for _ in range(nb)])
  return dict(speeds=speeds, contacts=contacts, poses=poses, dirs=dirs)


def run_case(ref, ctor, steps, rng):
  """One Simulator: reset(urdf) then the listed step calls.  steps: dicts with the script parameters + step kwargs."""
  W.__init__()
  W.order, W.loaded, W.placed = [], [], []
  sim = ref.Simulator(**ctor)
  sim.reset('rock_0')
  out = dict(ctor=ctor, setup=list(W.setup), max_step_count=int(sim._max_step_count), calls=[])
  assert W.log == [9] and sim.has_new_object and not sim.has_new_object
  for i, st in enumerate(steps):
    has_new = sim._new is not None
    nb = len(W.order) + (1 if has_new else 0)
    W.script = make_script(rng, nb, st['K'], st['end_smooth'], st['end_settle'], contacts_at_end=st.get('contacts_at_end', 3),
                           fast_old_until=st.get('fast_old_until'))
    W.k, W.log, W.w_out = 0, [], []
    pos = tuple(float(v) for v in rng.uniform(0.1, 0.4, size=3))
    orn = tuple(float(v) for v in unit(rng.normal(size=4)))
    raised = 0
    try:
      sim(position=pos, orientation=orn, urdf=st.get('next'), smooth_placing=st['smooth'])
    except RuntimeError as e:
      raised = 1
      assert 'Maximum number of simulator steps' in str(e)
    log = list(W.log)
    rec = dict(has_new=int(has_new), n_before=nb - (1 if has_new else 0), smooth=int(st['smooth']), raised=raised,
               speeds=W.script['speeds'], contacts=W.script['contacts'], poses=W.script['poses'], log=np.asarray(log, np.int32),
               asked_pose=np.asarray(pos + orn, np.float64), steps_taken=W.k)
    if not raised:
      rec['n_steps'] = np.asarray(sim.n_steps, np.int64)
      rec['place_poses'] = np.asarray([list(p) + list(o) for p, o in sim._place_poses], np.float64)
      rec['final_poses'] = np.asarray([list(p) + list(o) for p, o in sim.poses], np.float64)
      W.w_out = []
      rec['distances_from_place'] = np.asarray(sim.distances_from_place, np.float64).reshape(-1, 2)
      rec['w_handed_out'] = np.asarray(W.w_out, np.float64)
      assert sim.distances_from_place is sim._distances_from_place          # cached until the next step (simulator.py:116-117)
    out['calls'].append(rec)
    if raised:
      break
  return out


def main():
  ref = load_reference()
  assert ref.MAX_STEP_TIME == 300
  rng = np.random.RandomState(20240611)
  cases = []
  base = dict(time_step=0.01, gravity=9.8, spawn_position=(0, 0, 0.5), spawn_orientation=(0, 0, 0, 1), velocity_threshold=0.01)
  # A: Stack-v0's flow — smooth placing ended by three contacts, then the settle loop; four rocks, the last call without a next
  cases.append(run_case(ref, dict(base), [
    dict(K=40, end_smooth=5, end_settle=17, smooth=True, next='rock_1'),
    dict(K=40, end_smooth=1, end_settle=9, smooth=True, next='rock_2'),      # in touch at once: S_a = 1 + 1
    dict(K=40, end_smooth=0, end_settle=0, smooth=True, next='rock_3'),      # three contacts and at rest after _place: (1, 0)
    dict(K=60, end_smooth=12, end_settle=12, smooth=True, next=None),        # at rest the moment it has three contacts
    dict(K=20, end_smooth=0, end_settle=6, smooth=True, next=None),          # nothing waiting: no _place, loops still run
  ], rng))
  # B: smooth placing ended by the STOP criterion (never three contacts)
  cases.append(run_case(ref, dict(base), [
    dict(K=40, end_smooth=10 ** 6, end_settle=7, smooth=True, contacts_at_end=2, next='rock_1'),
    dict(K=40, end_smooth=10 ** 6, end_settle=11, smooth=True, contacts_at_end=0, next=None),
  ], rng))
  # C: without smooth placing (the reference's `smooth_placing=False`): no contact query, no velocity reset
  cases.append(run_case(ref, dict(base, time_step=0.0125), [
    dict(K=40, end_smooth=3, end_settle=8, smooth=False, next='rock_1'),
    dict(K=40, end_smooth=3, end_settle=0, smooth=False, next='rock_2'),
    dict(K=40, end_smooth=3, end_settle=21, smooth=False, next=None),
  ], rng))
  # D: an OLDER rock keeps moving after the newest has come to rest: the criterion asks every body, newest first
  cases.append(run_case(ref, dict(base), [
    dict(K=40, end_smooth=2, end_settle=4, smooth=True, next='rock_1'),
    dict(K=40, end_smooth=2, end_settle=5, smooth=True, next='rock_2', fast_old_until=14),
    dict(K=40, end_smooth=3, end_settle=6, smooth=True, next=None, fast_old_until=9),
  ], rng))
  # E: the step cap in the smooth-placing loop (time_step 30 s -> int(300 / 30) = 10 steps)
  cases.append(run_case(ref, dict(base, time_step=30.0), [
    dict(K=40, end_smooth=10 ** 6, end_settle=10 ** 6, smooth=True, contacts_at_end=0, next='rock_1'),
  ], rng))
  # F: the step cap in the settle loop (three contacts at once, never at rest); and one step short of it
  cases.append(run_case(ref, dict(base, time_step=25.0), [
    dict(K=40, end_smooth=0, end_settle=12, smooth=True, next='rock_1'),      # 1 + 11 = 12 steps = the cap: not raised
    dict(K=40, end_smooth=0, end_settle=10 ** 6, smooth=True, next=None),
  ], rng))
  # G: the cap with smooth placing off
  cases.append(run_case(ref, dict(base, time_step=50.0), [
    dict(K=40, end_smooth=0, end_settle=10 ** 6, smooth=False, next=None),
  ], rng))
  # the cap itself for the time steps the configurations use (simulator.py:46)
  caps = {}
  for ts in (0.01, 0.0125, 1 / 240., 0.005, 0.02, 1 / 60., 0.004, 0.001):
    caps[repr(ts)] = int(ref.Simulator(time_step=ts)._max_step_count)
  flat = {'n_cases': np.int64(len(cases)), 'cap_time_steps': np.asarray([float(k) for k in caps], np.float64),
          'cap_values': np.asarray(list(caps.values()), np.int64)}
  for ci, c in enumerate(cases):
    p = 'c{}_'.format(ci)
    flat[p + 'time_step'] = np.float64(c['ctor']['time_step'])
    flat[p + 'velocity_threshold'] = np.float64(c['ctor']['velocity_threshold'])
    flat[p + 'max_step_count'] = np.int64(c['max_step_count'])
    flat[p + 'setup'] = np.asarray(c['setup'])
    flat[p + 'n_calls'] = np.int64(len(c['calls']))
    for si, r in enumerate(c['calls']):
      for k, v in r.items():
        flat['{}s{}_{}'.format(p, si, k)] = np.asarray(v)
  np.savez_compressed(OUT, **flat)
  print('wrote', OUT, os.path.getsize(OUT), 'bytes;', len(cases), 'cases,', sum(len(c['calls']) for c in cases), 'step calls')
  for ci, c in enumerate(cases):
    for si, r in enumerate(c['calls']):
      print(ci, si, 'raised' if r['raised'] else tuple(r['n_steps']), 'log', len(r['log']), 'steps', r['steps_taken'])


if __name__ == '__main__':
  main()
