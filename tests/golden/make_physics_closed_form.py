"""Closed-form expectations for the rigid-body rows P1-P5 -> physics_closed_form.json.

The reference's physics lives in the unpinned pybullet wheel, so these analytic cases are the only
pins the settle solver can have: every number below follows from the reference's own parameters (env.py:28-51,
simulator.py:143-188, :212-224) and from the documented solver definition by pencil-and-paper
mechanics, evaluated here in float64.  Nothing under oracle/ or stackrl_amd/ is imported: the oracle AND the HIP kernels
are checked against this file (tests/test_physics_closed_form.py).

  python tests/golden/make_physics_closed_form.py
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# reference parameters
G = 9.8                  # gravity, env.py:35 -> simulator.py:166
DT = 0.01                # sim_time_step, env.py:34 -> simulator.py:143
DAMPING = 0.04           # pybullet default linear damping of a body (changeDynamics default)
MARGIN = 0.001           # pybullet URDF collision margin
SLOP = 1e-5              # pybullet server m_linearSlop
MU_GROUND = 0.6 * 0.5    # lateral_friction of a rock (template.urdf) x Bullet's default body friction of the ground
# cuboid `0_*` of the reference pool (envs/data/generated/0_*.obj): half extents
HX, HY, HZ = 0.05357143, 0.02678571, 0.01785714
RADIUS = math.sqrt(HX * HX + HY * HY + HZ * HZ)   # = object_max_dimension / 2 = 0.0625
BREAK = 0.02 * RADIUS    # Bullet's contact breaking threshold gContactBreakingThreshold x angular motion disc


def free_fall(z0, n):
  """stepSimulation on a free body from rest: damping (btRigidBody::applyDamping: v *= (1 - d)^dt), gravity impulse,
  symplectic Euler."""
  d = (1.0 - DAMPING) ** DT
  v, z, out = 0.0, z0, []
  for _ in range(n):
    v = v * d - G * DT
    z = z + v * DT
    out.append(z)
  return out


def slide(v0):
  """A flat cuboid sliding on the ground along a friction axis: per sub-step the friction row removes at most
  mu g dt of tangential speed (normal impulse = m g dt), after damping; symplectic Euler."""
  d = (1.0 - DAMPING) ** DT
  v, x, n = v0, 0.0, 0
  while v > 0.0:
    v = max(v * d - MU_GROUND * G * DT, 0.0)
    x += v * DT
    n += 1
  return x, n


def smooth_placing_steps(h):
  """S_a (simulator.py:212-224) for a flat cuboid released with its underside h above the ground: the place sub-step
  and every smooth-placing sub-step start from zero velocity and move the body down by g dt^2; the ground manifold of a
  sub-step is built from the pose at its start and holds the four bottom vertices once underside - margin < breaking
  threshold; _drop (>= 3 contact points, simulator.py:337-341) is evaluated after the sub-step."""
  k = 0   # sub-steps done so far; the underside is at h - k g dt^2 at the start of sub-step k (k = 0: _place's)
  while not (h - k * G * DT * DT - MARGIN < BREAK):
    k += 1
  return k + 1


def main():
  v0 = 1.0
  dist, nslide = slide(v0)
  out = {
    'source': 'tests/golden/make_physics_closed_form.py (float64 closed forms; no build or reference code involved)',
    'params': dict(gravity=G, dt=DT, damping=DAMPING, margin=MARGIN, linear_slop=SLOP, mu_ground=MU_GROUND,
                   cuboid_half_extents=[HX, HY, HZ], breaking_threshold=BREAK),
    # a cuboid lying flat on the ground at rest: centre height = half height + margin - slop (the normal row's target
    # is -(distance + slop) erp / dt, which vanishes at distance = -slop)
    'rest_z_one': HZ + MARGIN - SLOP,
    # a second cuboid flat on the first: two margins of their own contact, one of the ground contact, two slops
    'rest_z_two': 3 * HZ + 3 * MARGIN - 2 * SLOP,
    'rest_tol': 2e-5,
    'free_fall': dict(z0=0.3, z=free_fall(0.3, 20), tol=1e-6),
    'slide': dict(v0=v0, distance=dist, substeps=nslide, continuum=v0 * v0 / (2 * MU_GROUND * G), tol_rel=0.01),
    's_a': {'%.4f' % h: smooth_placing_steps(h) for h in (0.0, 0.002, 0.01, 0.0366, 0.05)},
  }
  with open(os.path.join(HERE, 'physics_closed_form.json'), 'w') as f:
    json.dump(out, f, indent=1)
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main()
