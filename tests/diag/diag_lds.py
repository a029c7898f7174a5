#!/usr/bin/env python3
"""Does an env step depend on what other kernels left in a CU's LDS?  One handle of B envs stepped in lock step with the
oracle; before every step an LDS poisoner (tools/experiments/lds_poison.hip) fills all 160 KB of every CU with a pattern."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stackrl_amd import assets, env as envs
from stackrl_amd.config import StackConfig
from oracle import oracle
B, L = int(sys.argv[1]), int(sys.argv[2])
patterns = [None if x == 'none' else int(x, 0) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [None]
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools')
KIND = os.environ.get('POISON', 'lds')      # lds | vgpr (tools/experiments/vgpr_poison.hip: every vector register of every SIMD)
P = ctypes.CDLL(os.path.join(here, 'experiments', KIND + '_poison.so'))
poison = getattr(P, KIND + '_poison')
poison.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device='cuda')
pool = assets.default_pool()
for pat in patterns:
  e = envs.VecStackEnv(n_parallel=B, seed=5, pool=pool, episode_length=L, side_stream=bool(int(os.environ.get('SIDE', '0'))))
  o = oracle.OracleEnv(StackConfig(n_envs=B, episode_length=L), pool, seed=5)
  e.reset()(); o.reset()
  bad_total = 0
  for t in range(2 * (L + 1)):
    a = e.sample(); ao = o.sample()
    assert np.array_equal(a.cpu().numpy(), ao)
    if pat is not None:
      st = e._side if e._side is not None else torch.cuda.current_stream()
      poison(pat, 1024 if KIND == 'lds' else 8192, ctypes.c_void_p(st.cuda_stream), ctypes.c_void_p(sink.data_ptr()))
    (om, oo), r, d = e.step(a)()
    (omo, ooo), ro, do = o.step(ao)
    om = om.cpu().numpy()
    badm = (om != omo).reshape(B, -1).sum(1)
    badr = (r.cpu().numpy().view(np.uint32) != ro.view(np.uint32))
    bado = (oo.cpu().numpy() != ooo).reshape(B, -1).sum(1)
    if badm.any() or badr.any() or bado.any():
      bad_total += int((badm > 0).sum())
      idx = np.nonzero(badm)[0][:8]
      print('pattern', pat if pat is None else hex(pat), 'call', t, 'obs_map differs in', int((badm > 0).sum()), 'envs', idx.tolist(), 'pixels', badm[idx].tolist(),
            'reward differs in', int(badr.sum()), 'obs_obj in', int((bado > 0).sum()))
      i = idx[0]
      dd = np.nonzero((om[i] != omo[i]).any(-1))
      print('   env', i, 'rows', dd[0].min(), dd[0].max(), 'cols', dd[1].min(), dd[1].max(), 'hip', om[i][dd][:4].tolist(), 'oracle', omo[i][dd][:4].tolist())
  print('pattern', pat if pat is None else hex(pat), 'done: envs-with-mismatch total', bad_total)
  e.close()
