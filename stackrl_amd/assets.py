"""Rock mesh pool: reader for the reference's asset format and a synthetic generator.

* Asset format (`stackrl/envs/data/generated/<name>.obj` + `.urdf`, written by
  `stackrl/envs/data/generator.py:242-266` from `template.urdf:1-26`): OBJ with only `v`/`f`
  records, URDF with `mass`, inertial `origin xyz` (centre of mass in the link frame) and
  `lateral_friction`.  `load_obj_urdf` reads exactly that subset.
* Synthetic rocks follow the `box` recipe of `generator.py:68-117` and `generate`
  (`:199-224`): cuboid with extents proportional to (1, 1/2, 1/3) inside a sphere of radius
  0.0625, truncated-normal vertex noise, 3 subdivisions with halving noise, convex hull, centre
  on the centre of mass, scale to fit, align the oriented bounding box, rotate 90 deg about Y,
  density U(2200, 2600).  trimesh is not available: hull/OBB/mass are restated on scipy/numpy.

A pool is a flat, C-ABI-ready bundle (`MeshPool`) handed to `srl_load_meshes`.
"""
import dataclasses
import glob
import os
import re

import numpy as np

try:
  from scipy.spatial import ConvexHull
  from scipy import stats
except ImportError:  # pragma: no cover - scipy is present in the image
  ConvexHull = None
  stats = None

from stackrl_amd.config import MAX_TRIS, MAX_VERTS


@dataclasses.dataclass
class MeshPool:
  verts: np.ndarray      # float32 [total_verts, 3], URDF link frame
  vert_off: np.ndarray   # int32 [n+1]
  tris: np.ndarray       # int32 [total_tris, 3], per-mesh local indices, outward CCW
  tri_off: np.ndarray    # int32 [n+1]
  mass_com: np.ndarray   # float32 [n, 4] = mass, com xyz (link frame)
  names: list

  def __len__(self):
    return len(self.vert_off) - 1

  def mesh(self, i):
    v = self.verts[self.vert_off[i]:self.vert_off[i + 1]]
    t = self.tris[self.tri_off[i]:self.tri_off[i + 1]]
    return v, t, self.mass_com[i]

  def subset(self, idx):
    return pack([self.mesh(i) for i in idx], [self.names[i] for i in idx])

  def select(self, urdfs):
    """The sub-pool whose file names match `urdfs`, the glob pattern (or list of patterns / irregularity numbers) of
    `StackEnv(urdfs=...)` (env.py:92-103: `data.generated(name=...)`), e.g. '[5-9]?' or '5?' or 50."""
    import fnmatch
    pats = urdfs if isinstance(urdfs, (list, tuple)) else [urdfs]
    pats = [str(p) if '_' in str(p) else str(p) + '_*' for p in pats]      # file names are '<irregularity>_<index>'
    idx = [i for i, n in enumerate(self.names) if any(fnmatch.fnmatch(n, p) for p in pats)]
    if not idx:
      raise ValueError('no mesh of the pool matches {}'.format(urdfs))
    return self.subset(idx)

  def save(self, path):
    np.savez_compressed(path, verts=self.verts, vert_off=self.vert_off, tris=self.tris,
                        tri_off=self.tri_off, mass_com=self.mass_com,
                        names=np.array(self.names))

  @classmethod
  def load(cls, path):
    z = np.load(path, allow_pickle=False)
    return cls(z['verts'], z['vert_off'], z['tris'], z['tri_off'], z['mass_com'],
               [str(s) for s in z['names']])


def pack(meshes, names=None):
  vo, to = [0], [0]
  for v, t, _ in meshes:
    if len(v) > MAX_VERTS or len(t) > MAX_TRIS:
      raise ValueError('mesh exceeds {} vertices / {} triangles'.format(MAX_VERTS, MAX_TRIS))
    vo.append(vo[-1] + len(v))
    to.append(to[-1] + len(t))
  return MeshPool(
    np.ascontiguousarray(np.concatenate([m[0] for m in meshes]), dtype=np.float32),
    np.array(vo, dtype=np.int32),
    np.ascontiguousarray(np.concatenate([m[1] for m in meshes]), dtype=np.int32),
    np.array(to, dtype=np.int32),
    np.ascontiguousarray(np.stack([m[2] for m in meshes]), dtype=np.float32),
    list(names) if names is not None else [str(i) for i in range(len(meshes))],
  )


# ----------------------------------------------------------------------------- asset reader
_URDF_MASS = re.compile(r'<mass\s+value\s*=\s*"([^"]+)"')
_URDF_ORIGIN = re.compile(r'<inertial>.*?<origin\s+xyz="([^"]+)"', re.S)
_URDF_FRICTION = re.compile(r'<lateral_friction\s+value="([^"]+)"')


def load_obj(path):
  v, f = [], []
  with open(path) as fh:
    for line in fh:
      if line.startswith('v '):
        v.append([float(x) for x in line.split()[1:4]])
      elif line.startswith('f '):
        f.append([int(x.split('/')[0]) - 1 for x in line.split()[1:4]])
  return np.array(v, dtype=np.float32), np.array(f, dtype=np.int32)


def load_urdf(path):
  with open(path) as fh:
    txt = fh.read()
  mass = float(_URDF_MASS.search(txt).group(1))
  com = [float(x) for x in _URDF_ORIGIN.search(txt).group(1).split()]
  fr = _URDF_FRICTION.search(txt)
  return mass, com, float(fr.group(1)) if fr else 0.5


def load_obj_urdf(urdf_path):
  """One rock from `<name>.urdf` (+ the `<name>.obj` next to it)."""
  mass, com, _ = load_urdf(urdf_path)
  v, t = load_obj(os.path.splitext(urdf_path)[0] + '.obj')
  return v, t, np.array([mass] + list(com), dtype=np.float32)


def load_directory(directory, pattern='[5-9]?'):
  """Pool from a directory of reference assets; `pattern` as `data.generated(name=...)`
  (`stackrl/envs/data/__init__.py:39-83`: glob `<name>_*.urdf`)."""
  files = sorted(glob.glob(os.path.join(directory, '{}_*.urdf'.format(pattern))))
  if not files:
    raise AssertionError('List of object descriptor files is empty.')  # env.py:103
  return pack([load_obj_urdf(f) for f in files],
              [os.path.splitext(os.path.basename(f))[0] for f in files])


# ----------------------------------------------------------------------------- geometry helpers
def hull_mesh(points):
  """Convex hull as (vertices, outward-CCW triangles)."""
  hull = ConvexHull(points)
  used = np.unique(hull.simplices)
  remap = -np.ones(len(points), dtype=np.int64)
  remap[used] = np.arange(len(used))
  v = points[used]
  t = remap[hull.simplices]
  # orient outward using the facet equations (normal . x + d <= 0 inside)
  a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
  n = np.cross(b - a, c - a)
  flip = np.einsum('ij,ij->i', n, hull.equations[:, :3]) < 0
  t[flip] = t[flip][:, [0, 2, 1]]
  return v, t


def mass_properties(v, t):
  """Volume and centre of mass of a closed triangle mesh (signed tetrahedra)."""
  a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
  vol6 = np.einsum('ij,ij->i', a, np.cross(b, c))
  volume = vol6.sum() / 6.0
  com = ((a + b + c) * vol6[:, None]).sum(0) / (24.0 * volume)
  return volume, com


def _min_area_rect(p2):
  """Minimum-area enclosing rectangle of 2-D points: (area, angle)."""
  h = ConvexHull(p2)
  hp = p2[h.vertices]
  e = np.roll(hp, -1, axis=0) - hp
  ang = np.unique(np.mod(np.arctan2(e[:, 1], e[:, 0]), np.pi / 2))
  c, s = np.cos(ang), np.sin(ang)
  x = hp[:, 0][None] * c[:, None] + hp[:, 1][None] * s[:, None]
  y = -hp[:, 0][None] * s[:, None] + hp[:, 1][None] * c[:, None]
  area = (x.max(1) - x.min(1)) * (y.max(1) - y.min(1))
  k = int(np.argmin(area))
  return area[k], ang[k]


def oriented_bounds(v, t):
  """Minimum-volume oriented bounding box over hull-face directions (what trimesh's
  `bounding_box_oriented` / `apply_obb` search): returns (R, centre, extents) with rows of R the
  box axes, extents sorted as the axes are."""
  a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
  n = np.cross(b - a, c - a)
  n /= np.linalg.norm(n, axis=1, keepdims=True)
  # unique directions up to sign
  n = n * np.where(n[:, [np.argmax(np.abs(n).sum(0))]] < 0, -1, 1)
  n = np.unique(np.round(n, 6), axis=0)
  best = None
  for z in n:
    z = z / np.linalg.norm(z)
    x = np.cross(z, [1., 0, 0] if abs(z[0]) < 0.9 else [0, 1., 0])
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    p2 = np.stack([v @ x, v @ y], axis=1)
    hgt = v @ z
    area, ang = _min_area_rect(p2)
    vol = area * (hgt.max() - hgt.min())
    if best is None or vol < best[0]:
      ca, sa = np.cos(ang), np.sin(ang)
      best = (vol, np.stack([ca * x + sa * y, -sa * x + ca * y, z]))
  R = best[1]
  pr = v @ R.T
  lo, hi = pr.min(0), pr.max(0)
  ext = hi - lo
  order = np.argsort(ext)   # shortest extent on x, longest on z (the 90 deg turn about Y in
                            # generate_rock then puts the longest on x, as in the reference pool)
  R = R[order]
  if np.linalg.det(R) < 0:
    R[2] = -R[2]
  pr = v @ R.T
  lo, hi = pr.min(0), pr.max(0)
  return R, R.T @ ((lo + hi) / 2), hi - lo


def _subdivide(v, t):
  """Replace each triangle with four (midpoint subdivision); new vertices appended."""
  e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1)
  ue, inv = np.unique(e, axis=0, return_inverse=True)
  mid = (v[ue[:, 0]] + v[ue[:, 1]]) / 2
  nv = len(v)
  m = inv.reshape(3, -1) + nv
  a, b, c = t[:, 0], t[:, 1], t[:, 2]
  ab, bc, ca = m[0], m[1], m[2]
  t2 = np.concatenate([
    np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1),
    np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)])
  return np.concatenate([v, mid]), t2


_BOX_TRIS = np.array([
  [0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1],
  [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], dtype=np.int64)


def box_rock(rng, radius=0.0625, irregularity=0., extents=(1, 1 / 2, 1 / 3), subdivisions=3):
  """`generator.box` (generator.py:68-117) restated."""
  ext = np.array(extents, dtype=np.float64) * 2 * radius / np.linalg.norm(extents)
  v = np.array([[x, y, z] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)]) * ext
  t = _BOX_TRIS.copy()

  def noise(shape, scale):
    return stats.truncnorm.rvs(-1 / irregularity, 1 / irregularity, loc=0, scale=scale,
                               size=shape, random_state=rng)

  if irregularity > 0:
    v = v + noise(v.shape, irregularity * radius)
  for i in range(subdivisions):
    nv = len(v)
    v, t = _subdivide(v, t)
    if irregularity > 0:
      v[nv:] += noise(v[nv:].shape, irregularity * radius * 2 ** (-(i + 1)))
  v, t = hull_mesh(v)
  _, com = mass_properties(v, t)
  v = v - com
  # generator.py:114-116 scales by 2*radius/max(OBB extents); the committed reference pool
  # (data/generated/[5-9]?_*) instead fits the bounding SPHERE of radius `radius` about the centre
  # of mass (measured: mean max-vertex-radius 0.0635, extents mean (0.105, 0.072, 0.050)), so the
  # pool statistics SURVEY.md section 8d asks to match are reproduced with the sphere fit.
  factor = radius / np.linalg.norm(v, axis=1).max()
  if factor < 1:
    v = v * factor
  return v, t


def generate_rock(rng, irregularity, density=(2200., 2600.)):
  """One rock as `generator.generate` writes it (generator.py:199-266)."""
  v, t = box_rock(rng, irregularity=irregularity)
  R, centre, _ = oriented_bounds(v, t)        # mesh.apply_obb()
  v = (v - centre) @ R.T
  v = v @ np.array([[0., 0, -1], [0, 1, 0], [1, 0, 0]])  # rotation_matrix(pi/2, [0,1,0]) applied to rows
  v, t = hull_mesh(v)
  rho = rng.uniform(density[0], density[1])
  volume, com = mass_properties(v, t)
  return (v.astype(np.float32), t.astype(np.int32),
          np.array([rho * volume, com[0], com[1], com[2]], dtype=np.float32))


def generate_pool(n=5000, seed=11, irregularities=None):
  """Synthetic stand-in for the `[5-9]?` families (10 families, irregularity 0.50..0.95,
  `n // 10` rocks each); generator seed 11."""
  if irregularities is None:
    irregularities = [i / 100. for i in range(50, 100, 5)]
  rng = np.random.default_rng(seed)
  meshes, names = [], []
  per = max(1, n // len(irregularities))
  for irr in irregularities:
    for k in range(per):
      if len(meshes) >= n:
        break
      meshes.append(generate_rock(rng, irr))
      names.append('{}_{:03d}'.format(int(round(irr * 100)), k))
  return pack(meshes, names)


def cuboid(extents=(0.10714286, 0.05357143, 0.03571429), density=2400.):
  """Axis-aligned cuboid like the reference's `0_*.obj` (8 vertices, 12 triangles)."""
  ext = np.array(extents, dtype=np.float64)
  v = np.array([[x, y, z] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)]) * ext
  return (v.astype(np.float32), _BOX_TRIS.astype(np.int32),
          np.array([density * ext.prod(), 0, 0, 0], dtype=np.float32))


def default_pool(n=5000, seed=11, cache_dir=None):
  """The bench/test pool (5,000 synthetic rocks, generator seed 11).  The default pool ships as data
  (`stackrl_amd/data/pool_5000_11.npz`, written by this very function); other sizes/seeds are generated once and
  cached next to the package (git-ignored)."""
  here = os.path.dirname(os.path.abspath(__file__))
  shipped = os.path.join(here, 'data', 'pool_{}_{}.npz'.format(n, seed))
  if os.path.isfile(shipped):
    return MeshPool.load(shipped)
  cache_dir = cache_dir or os.path.join(here, '_cache')
  path = os.path.join(cache_dir, 'pool_{}_{}.npz'.format(n, seed))
  if os.path.isfile(path):
    try:
      return MeshPool.load(path)
    except Exception:  # corrupted cache: regenerate
      pass
  pool = generate_pool(n, seed)
  try:
    os.makedirs(cache_dir, exist_ok=True)
    tmp = path + '.tmp.{}.npz'.format(os.getpid())
    pool.save(tmp)
    os.replace(tmp, path)
  except OSError:
    pass
  return pool
