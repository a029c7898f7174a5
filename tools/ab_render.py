#!/usr/bin/env python3
"""A/B timing of render-kernel build variants on one box: each variant (a set of extra -D flags) is compiled into
gpurun_out/, then the variants are timed in turn, twice (ABAB), on the explicit-pose micro-benchmark.
usage: ab_render.py "" "-DSRL_X=1" "-DSRL_X=2 -DSRL_Y" ...      (compile on the box, then time)
       ab_render.py --libs DIR                                  (time every lib*.so of DIR, e.g. cross-compiled variants
                                                                 of different source revisions; DIR must travel: not gpurun_out/)"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stackrl_amd import build as B

def run(so):
  B.LIB = so
  import numpy as np, torch
  from stackrl_amd import assets, env as envs
  nB = 1024
  pool = assets.default_pool()
  g = envs.VecStackEnv(n_parallel=nB, seed=11, pool=pool, block=True, episode_length=8)
  rng = np.random.RandomState(0)
  res = []
  for nbv in [0, 4, 8]:
    poses = np.zeros((nB, 32, 7), np.float32); mesh = np.zeros((nB, 32), np.int32)
    for b in range(nbv):
      q = rng.normal(size=(nB, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
      poses[:, b, 0] = rng.uniform(0.05, 0.45, nB); poses[:, b, 1] = rng.uniform(0.05, 0.45, nB); poses[:, b, 2] = rng.uniform(0.03, 0.12, nB)
      poses[:, b, 3:] = q
      mesh[:, b] = rng.randint(len(pool), size=nB)
    P = torch.from_numpy(poses).cuda(); M = torch.from_numpy(mesh).cuda(); N = torch.full((nB,), nbv, dtype=torch.int32).cuda()
    out = torch.empty((nB, 128, 128), dtype=torch.float32, device='cuda')
    for _ in range(10): g.render_heightmap(P, M, N, out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): g.render_heightmap(P, M, N, out)
    e1.record(); torch.cuda.synchronize()
    res.append('nb%d %.1f' % (nbv, e0.elapsed_time(e1) / 200 * 1e3))
  # the step path: full observation + reward (the bench's launch), from the library's own HIP events
  g.reset()
  for _ in range(9): g.step(g.sample())
  g.kernel_times(); g.set_profiling(True)
  for _ in range(54): g.step(g.sample())
  torch.cuda.synchronize()
  ms, n = g.kernel_times()
  g.set_profiling(False)
  res.append('step-path %.2f us (+ stage %.2f us)' % (ms[1] / max(n[1], 1) * 1e3, ms[2] / max(n[2], 1) * 1e3))
  print('  '.join(res), flush=True)

if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == '--run':
    run(sys.argv[2]); sys.exit(0)
  variants = sys.argv[1:]
  sos = []
  if variants and variants[0] == '--libs':
    import glob
    sos = sorted(x for x in glob.glob(os.path.join(os.path.abspath(variants[1]), 'lib*.so')) if 'qnet' not in x)
    variants = [os.path.basename(x) for x in sos]
  for k, v in enumerate([] if sos else variants):
    so = os.path.join(ROOT, 'gpurun_out', 'libstackrl_ab%d.so' % k)
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + B.FLAGS + v.split() + [os.path.join(B.CSRC, 'stackrl_hip.hip'), '-o', so])
    sos.append(so)
  for rep in range(2):
    for k, so in enumerate(sos):
      out = subprocess.run([sys.executable, os.path.abspath(__file__), '--run', so], check=True, stdout=subprocess.PIPE, universal_newlines=True).stdout
      print('[%d] %-40s %s' % (k, variants[k] or '(default)', out.strip().splitlines()[-1]), flush=True)
