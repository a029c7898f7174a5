"""Asset reader + synthetic generator (SURVEY.md section 8d "Synthetic rocks")."""
import numpy as np

from stackrl_amd import assets


def test_reference_subset_fixture(ref_pool):
  assert len(ref_pool) == 69
  nv = np.diff(ref_pool.vert_off); nt = np.diff(ref_pool.tri_off)
  assert nv.min() == 8 and nv.max() <= 70 and nt.max() <= 136
  for i in range(len(ref_pool)):
    v, t, mc = ref_pool.mesh(i)
    vol, com = assets.mass_properties(v.astype(np.float64), t)
    assert vol > 0                                         # outward orientation
    assert len(v) - 3 * len(t) // 2 + len(t) == 2          # closed genus-0 surface (Euler)
    assert np.abs(com - mc[1:]).max() < 1e-4               # URDF inertial origin == centre of mass
    assert 2200 * vol * 0.999 <= mc[0] <= 2600 * vol * 1.001   # density U(2200,2600), generator.py:128


def test_synthetic_pool_matches_reference_statistics():
  pool = assets.generate_pool(60, seed=11)
  nv = np.diff(pool.vert_off); nt = np.diff(pool.tri_off)
  assert 24 <= nv.min() and nv.max() <= 80 and 38 <= nv.mean() <= 52      # reference: 24 / 44.7 / 70
  assert 70 <= nt.mean() <= 100                                           # reference: 85.5
  ext = np.array([pool.mesh(i)[0].max(0) - pool.mesh(i)[0].min(0) for i in range(len(pool))])
  assert np.allclose(ext.mean(0), [0.105, 0.072, 0.050], atol=0.012)      # reference extents mean
  assert 0.30 <= pool.mass_com[:, 0].mean() <= 0.55                       # reference mean 0.436 kg
  again = assets.generate_pool(60, seed=11)
  assert np.array_equal(pool.verts, again.verts) and np.array_equal(pool.tris, again.tris)


def test_obj_urdf_roundtrip(tmp_path, ref_pool):
  v, t, mc = ref_pool.mesh(3)
  with open(tmp_path / 'r_0.obj', 'w') as f:
    f.write('# test\n')
    for p in v:
      f.write('v {:.8f} {:.8f} {:.8f}\n'.format(*p))
    for tri in t:
      f.write('f {} {} {}\n'.format(*(tri + 1)))
  with open(tmp_path / 'r_0.urdf', 'w') as f:
    f.write('<robot name="r_0"><link name="link"><contact><lateral_friction value="0.6"/></contact>'
            '<inertial>\n<origin xyz="{} {} {}" rpy="0 0 0"/>\n<mass value = "{}"/>'
            '<inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1" /></inertial></link></robot>'
            .format(mc[1], mc[2], mc[3], mc[0]))
  pool = assets.load_directory(str(tmp_path), 'r')
  v2, t2, mc2 = pool.mesh(0)
  assert np.allclose(v2, v, atol=1e-7) and np.array_equal(t2, t) and np.allclose(mc2, mc, rtol=1e-6)


def test_shipped_default_pool_is_the_generators_output():
  """The 5,000-rock pool shipped as data is what `generate_pool(seed=11)` produces (first family re-generated)."""
  pool = assets.default_pool()
  assert len(pool) == 5000
  head = assets.generate_pool(500, seed=11, irregularities=[0.5])
  n = head.vert_off[20]
  assert np.array_equal(pool.vert_off[:21], head.vert_off[:21])
  assert np.array_equal(pool.verts[:n], head.verts[:n])
