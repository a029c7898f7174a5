#!/usr/bin/env python3
"""Copy a small subset of the reference's rock ASSETS (data, MIT-licensed) into a fixture.

Runs only in the build container.  64 rocks of the Stack-v0 pool (`[5-9]?`, every 78th file of
the sorted list) plus the five `0_*` cuboids, read with the product's own OBJ/URDF reader
(`stackrl_amd.assets`), written to tests/golden/ref_rocks.npz (vertices, triangles, mass, COM).
"""
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from stackrl_amd import assets  # noqa: E402

GEN = '/root/reference/stackrl/envs/data/generated'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_rocks.npz')


def main():
  files = sorted(glob.glob(os.path.join(GEN, '[5-9]?_*.urdf')))
  assert len(files) == 5000, len(files)
  pick = files[::78][:64]
  pick += sorted(glob.glob(os.path.join(GEN, '0_*.urdf')))
  pool = assets.pack([assets.load_obj_urdf(f) for f in pick],
                     [os.path.splitext(os.path.basename(f))[0] for f in pick])
  pool.save(OUT)
  print('wrote', OUT, len(pool), 'meshes', os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
  main()
