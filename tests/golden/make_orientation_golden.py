#!/usr/bin/env python3
"""Golden vectors for the Stack-v2 orientation list, from the reference's own Observer (numpy-only file).

Runs ONLY in the build container (needs /root/reference).  `Observer(..., orientation_freedom=k)`
(observer.py:127-140, :278-293, :392-421) is driven with a stub simulator whose transform helpers implement the
published pybullet semantics with numpy (`getQuaternionFromEuler` for a pure yaw, `multiplyTransforms`,
`invertTransform`; quaternions xyzw).  Recorded per k: the number of object maps one observation holds, the
orientation `pose(pixel, index)` returns for every index, and the camera up-vector each map was rendered with.
What this pins is the reference's *structure* — orientation i = inverse of the yaw i * 2 pi / 2^k, in index order —
the trigonometry itself is the stub's.  Output: orientation_golden.npz (data only)."""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True   # the reference tree is read-only: no __pycache__ beside its files

import numpy as np

REF = '/root/reference/stackrl/envs/stack/observer.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'orientation_golden.npz')


def qmul(a, b):
  ax, ay, az, aw = a; bx, by, bz, bw = b
  return (aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
          aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz)


def qrot(q, v):
  x, y, z, _ = qmul(qmul(q, (v[0], v[1], v[2], 0.)), (-q[0], -q[1], -q[2], q[3]))
  return (x, y, z)


class StubSim(object):
  def __init__(self, spawn):
    self.new_pose = (tuple(spawn), (0., 0., 0., 1.))
    self.has_new_object = True
    self.ups = []

  def computeViewMatrix(self, **kw):
    self.ups.append(tuple(kw['cameraUpVector']))
    return ('view', kw)

  def computeProjectionMatrix(self, **kw):
    return ('proj', kw)

  def getQuaternionFromEuler(self, e):
    assert e[0] == 0 and e[1] == 0
    return (0., 0., float(np.sin(e[2] / 2)), float(np.cos(e[2] / 2)))

  def multiplyTransforms(self, pa, qa, pb, qb):
    return tuple(np.add(pa, qrot(qa, pb))), qmul(qa, qb)

  def invertTransform(self, p, q):
    qi = (-q[0], -q[1], -q[2], q[3])
    return tuple(-np.asarray(qrot(qi, p))), qi

  def getCameraImage(self, width, height, viewMatrix, projectionMatrix):
    return width, height, None, np.full((height, width), 0.5, 'float32'), None


def main():
  spec = importlib.util.spec_from_file_location('ref_observer', REF)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  out = {}
  for k in (1, 2, 3, 4):
    sim = StubSim((0., 0., 0.5))
    obs = mod.Observer(sim, overhead_resolution=128, object_resolution=32, pixel_size=0.125 / 32, max_z=0.375,
                       orientation_freedom=k)
    sim.ups = []
    obs()
    m, n = obs.state
    out['k%d_n_maps' % k] = np.array(len(n), 'int64')
    out['k%d_orientation' % k] = np.array([obs.pose((3, 4), index=i)['orientation'] for i in range(len(n))], 'float64')
    out['k%d_up' % k] = np.array(sim.ups[-len(n):], 'float64')
    out['k%d_position' % k] = np.array(obs.pose((3, 4), index=0)['position'], 'float64')
  np.savez_compressed(OUT, **out)
  print('wrote', OUT, {key: v.shape for key, v in out.items()})


if __name__ == '__main__':
  sys.exit(main())
