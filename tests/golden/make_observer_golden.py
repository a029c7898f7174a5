#!/usr/bin/env python3
"""Generate golden vectors from the reference's own Observer (numpy-only file).

Runs ONLY in the build container (needs /root/reference). It imports
`stackrl/envs/stack/observer.py` by file path (the package itself cannot be
imported: pybullet/gym/tensorflow are absent) with a stub simulator that
replays seeded synthetic depth buffers, and records

  * depth -> elevation for the overhead camera   (observer.py:259-260)
  * depth -> elevation + column flip, object cam (observer.py:274-277)
  * Observer.pose() for random (H, O, pixel)      (observer.py:392-421)
  * shape / size / max_z / pixel_to_xy / xy_to_pixel (observer.py:354-390)

The output `observer_golden.npz` holds data only (inputs + expected outputs).
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True   # the reference tree is read-only: no __pycache__ beside its files

import numpy as np

REF = '/root/reference/stackrl/envs/stack/observer.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'observer_golden.npz')


class StubSim(object):
  """Duck-typed pybullet facade (observer.py:41-49): replays depth buffers."""

  def __init__(self, spawn):
    self.new_pose = (tuple(spawn), (0., 0., 0., 1.))
    self.has_new_object = True
    self.queue = []

  def computeViewMatrix(self, **kw):
    return ('view', kw)

  def computeProjectionMatrix(self, **kw):
    return ('proj', kw)

  def getQuaternionFromEuler(self, e):
    return (0., 0., 0., 1.)

  def multiplyTransforms(self, pa, qa, pb, qb):
    return tuple(np.add(pa, pb)), qa

  def invertTransform(self, p, q):
    return tuple(-np.asarray(p)), q

  def getCameraImage(self, width, height, viewMatrix, projectionMatrix):
    d = self.queue.pop(0)
    assert d.shape == (height, width)
    return width, height, None, d, None


def main():
  spec = importlib.util.spec_from_file_location('ref_observer', REF)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)

  rng = np.random.RandomState(11)
  out = {}
  for tag, (H, h, omd, max_z) in {
      'a': (128, 32, 0.125, 0.375),   # Stack-v0 defaults (env.py:128-136)
      'b': (64, 16, 0.125, 0.375),    # resolution_factor=4 (BASELINE config 5)
  }.items():
    sim = StubSim((0., 0., max_z + omd))
    obs = mod.Observer(sim, overhead_resolution=H, object_resolution=h,
                       pixel_size=omd / h, max_z=max_z)
    n_case = 6
    d_over = rng.uniform(0, 1, size=(n_case, H, H)).astype('float32')
    d_obj = rng.uniform(0, 1, size=(n_case, h, h)).astype('float32')
    # exercise the end points and "empty" (d == 1) pixels
    d_over[0, :4] = 1.0
    d_over[0, 4:8] = 0.0
    d_obj[0, :2] = 1.0
    d_obj[0, 2:4] = 0.0
    d_obj[1:, :, :3] = 1.0
    d_obj[1:, :, -2:] = 1.0
    e_over, e_obj, pix, poses = [], [], [], []
    for k in range(n_case):
      sim.queue = [d_over[k], d_obj[k]]
      sim.has_new_object = True
      obs()
      m, n = obs.state
      e_over.append(np.array(m))
      e_obj.append(np.array(n))
      pk, qk = [], []
      for _ in range(16):
        u = int(rng.randint(0, H - h + 1))
        v = int(rng.randint(0, H - h + 1))
        p = obs.pose((u, v))['position']
        pk.append((u, v))
        qk.append([float(p[0]), float(p[1]), float(p[2])])
      pix.append(pk)
      poses.append(qk)
    out[tag + '_depth_over'] = d_over
    out[tag + '_depth_obj'] = d_obj
    out[tag + '_elev_over'] = np.stack(e_over)
    out[tag + '_elev_obj'] = np.stack(e_obj)
    out[tag + '_elev_over_dtype'] = np.array(str(e_over[0].dtype))
    out[tag + '_elev_obj_dtype'] = np.array(str(e_obj[0].dtype))
    out[tag + '_pose_pixel'] = np.array(pix, dtype='int64')
    out[tag + '_pose_xyz'] = np.array(poses, dtype='float64')
    out[tag + '_shape'] = np.array(obs.shape, dtype='int64')
    out[tag + '_size'] = np.array(obs.size, dtype='float64')
    out[tag + '_max_z'] = np.array(obs.max_z, dtype='float64')
    out[tag + '_p2xy'] = np.array(obs.pixel_to_xy((5, 77 % H)), dtype='float64')
    out[tag + '_xy2p'] = np.array(obs.xy_to_pixel((0.1234, 0.2345)), dtype='float64')
  np.savez_compressed(OUT, **out)
  print('wrote', OUT, {k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
  sys.exit(main())
