#!/usr/bin/env python3
"""Which concurrency makes an env step differ from the oracle?  B envs as `groups` handles on side streams, stepped in lock
step with the oracle; while the step is in flight the current stream runs `load`: none | matmul | qnet."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
if os.environ.get('SRL_DIAG_LIB'):      # run against another build of libstackrl_hip.so (diagnostic variants)
  from stackrl_amd import build as _b
  _b.LIB = os.path.abspath(os.environ['SRL_DIAG_LIB'])
  _b.stale = lambda: False
if os.environ.get('SRL_DIAG_QLIB'):     # ... and of libstackrl_qnet.so
  from stackrl_amd import build as _b
  _b.QLIB = os.path.abspath(os.environ['SRL_DIAG_QLIB'])
  _b.qstale = lambda: False
from stackrl_amd import assets, env as envs, nets, qops
from stackrl_amd.config import StackConfig
from oracle import oracle
B, L, groups, load = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
episodes = int(sys.argv[5]) if len(sys.argv) > 5 else 2
OFF = int(os.environ.get('ENV_OFFSET', '0'))
pool = assets.default_pool()
e = envs.make('Stack-v0', n_parallel=B, seed=5, pool=pool, episode_length=L, side_stream=True, env_index_offset=OFF, **({'groups': groups} if groups > 1 else {}))
o = oracle.OracleEnv(StackConfig(n_envs=B, episode_length=L, env_index_offset=OFF), pool, seed=5)
net = nets.DeepQSiamFCN(e.observation_spec, seed=2).cuda()
pol = qops.FusedPolicy(chunk=256, fast=True)
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
step = e.reset()(); o.reset()
filler = torch.randn(2048, 2048, device='cuda')
bad_total = 0
for t in range(episodes * (L + 1)):
  a = e.sample(); ao = o.sample()
  assert np.array_equal(a.cpu().numpy(), ao)
  pre = load.startswith('pre_')            # the load runs to completion BEFORE the step is launched
  if pre:
    ld = load[4:]
    if ld == 'pos':
      if pol._ff is None:
        pol._ff = qops.FastFeatures(net, dtype=torch.float32)
        C = torch.randn(256, 1, 97, 97, device='cuda')
      with torch.no_grad():
        for _ in range(40): pol._ff.pos(C)
    torch.cuda.synchronize()
  w = e.step(a, block=False)
  if load in ('ldsbusy', 'vgprbusy'):
    import ctypes
    kind = 'lds' if load == 'ldsbusy' else 'vgpr'
    Pz = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools', 'experiments', kind + '_poison.so'))
    fn = getattr(Pz, kind + '_poison'); fn.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for _ in range(10):
      fn(0xdeadbeef, 20000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None)
  if load == 'matmul':
    for _ in range(20):
      filler = (filler @ filler) * 1e-3
  elif load == 'qnet':
    pol(net, step[0], 1.0, gen)
  elif load in ('ff', 'xcorr', 'pos', 'head', 'thin', 'proj'):
    if pol._ff is None:
      pol._ff = qops.FastFeatures(net, dtype=torch.float32)
      with torch.no_grad():
        X, W = pol._ff((step[0][0][:256], step[0][1][:256]))
        C = qops.xcorr_forward(X, W)
        A = pol._ff.pos(C)
      U = torch.rand(256, device='cuda'); R = torch.randint(100, (256,), device='cuda')
    with torch.no_grad():
      for s0 in range(0, max(B, 512), 256):
        if load == 'ff':
          pol._ff((step[0][0][s0:s0 + 256], step[0][1][s0:s0 + 256]))
        elif load == 'xcorr':
          for _ in range(4): qops.xcorr_forward(X, W)
        elif load == 'pos':
          for _ in range(20): pol._ff.pos(C)
        elif load in ('thin', 'proj'):       # the two kernels of the position head, one at a time
          ff_ = pol._ff; posn = net.pos
          if not hasattr(ff_, '_dbuf'):
            ff_._dbuf = torch.zeros((256, 16, 112, 112), dtype=torch.float32, device='cuda').contiguous(memory_format=torch.channels_last)
            ff_._dcorr = C.reshape(256, 97, 97, 1).contiguous()
          for _ in range(20):
            if load == 'thin':
              qops.conv3x3_thin(ff_._dcorr, ff_._wt[posn[0]], ff_._w[posn[0]][1], out=ff_._dbuf)
            else:
              qops.conv3x3_relu_project(ff_._dbuf, ff_._wf[posn[2]], ff_._w[posn[2]][1], ff_._pos[0], ff_._pos[1], 97, 97)
        else:
          for _ in range(50): qops.policy_head(A, U, R, 0.5)
  step = w()
  (om, oo), r, d = step
  (omo, ooo), ro, do = o.step(ao)
  Hh, Oh, gh = e.maps(); Ho, Oo, go = o.maps()
  badH = (Hh.view(np.uint32) != Ho.view(np.uint32)).reshape(B, -1).sum(1)
  st_, so_ = e.state(), o.state()
  badP = (st_[0].view(np.uint32) != so_[0].view(np.uint32)).reshape(B, -1).any(1)
  badS = (st_[2] != so_[2]).reshape(B, -1).any(1)
  if badH.any() or badP.any():
    hp = np.nonzero((badH > 0) & ~badP)[0]
    print('call', t, ': H differs in', int((badH > 0).sum()), 'envs; poses in', int(badP.sum()), '; sub-step counts in', int(badS.sum()),
          '; H differs with EQUAL poses in', hp[:8].tolist(), [int(badH[i]) for i in hp[:8]])
    ip = np.nonzero(badP)[0][:4]
    for i in ip:
      d = np.abs(st_[0][i].astype(np.float64) - so_[0][i]).max()
      print('      env', int(i), 'max |pose diff|', d, 'substeps', st_[2][i].tolist(), so_[2][i].tolist(), 'nb', int(st_[1][i]))
  omh = om.cpu().numpy()
  badm = (omh != omo).reshape(B, -1).sum(1)
  badr = (r.cpu().numpy().view(np.uint32) != ro.view(np.uint32))
  if badm.any() or badr.any():
    bad_total += int((badm > 0).sum())
    idx = np.nonzero(badm)[0][:8]
    print('call', t, 'obs_map differs in', int((badm > 0).sum()), 'envs', idx.tolist(), 'pixels', badm[idx].tolist(), 'reward differs in', int(badr.sum()))
    if len(idx):
      i = idx[0]
      dd = np.nonzero((omh[i] != omo[i]).any(-1))
      print('   env', i, 'rows', dd[0].min(), dd[0].max(), 'cols', dd[1].min(), dd[1].max(), 'hip', omh[i][dd][:6].tolist(), 'oracle', omo[i][dd][:6].tolist())
    st = e.state(); so = o.state()
    for k in range(4):
      if not np.array_equal(st[k], so[k]):
        be = np.nonzero((st[k] != so[k]).reshape(B, -1).any(1))[0]
        print('   state', k, 'differs in envs', be[:8].tolist())
print('B', B, 'L', L, 'groups', groups, 'load', load, 'done: envs-with-mismatch total', bad_total)
e.close()
