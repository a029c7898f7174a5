#!/usr/bin/env python3
"""Is the render kernel's output the same while the Q-net's convolution kernels run on another stream?  The renderer hook on
fixed explicit poses (srl_render_heightmap), N launches alone and N launches under load, every output compared bit for bit
with the first.  SRL_DIAG_LIB selects the build of libstackrl_hip.so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('SRL_DIAG_LIB'):
  from stackrl_amd import build as _b
  _b.LIB = os.path.abspath(os.environ['SRL_DIAG_LIB']); _b.stale = lambda: False
import numpy as np, torch
from stackrl_amd import assets, env as envs, nets, qops
B, L, N = 1024, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 200
pool = assets.default_pool()
e = envs.VecStackEnv(n_parallel=B, seed=5, pool=pool, episode_length=L, side_stream=True)
e.reset()()
for _ in range(L):
  e.step(e.sample())()
st = e.state()
poses = torch.from_numpy(np.ascontiguousarray(st[0][:, :, :7])).cuda()
mesh = torch.from_numpy(np.ascontiguousarray(st[0][:, :, 7].astype(np.int32))).cuda()
nb = torch.from_numpy(st[1].astype(np.int32)).cuda()
net = nets.DeepQSiamFCN(e.observation_spec, seed=2).cuda()
ff = qops.FastFeatures(net, dtype=torch.float32)
C = torch.randn(256, 1, 97, 97, device='cuda')
xm = torch.randint(0, 256, (256, 128, 128, 2), device='cuda', dtype=torch.uint8)
xo = torch.randint(0, 256, (256, 32, 32, 1), device='cuda', dtype=torch.uint8)
with torch.no_grad(): ff.pos(C); ff((xm, xo))
ref = e.render_heightmap(poses, mesh, nb)
torch.cuda.synchronize()
ref = ref.clone()
outs = [torch.empty_like(ref) for _ in range(8)]
for load in ('none', 'pos', 'ff', 'none'):
  bad = 0; shown = 0
  for k in range(0, N, 8):
    with torch.no_grad():
      if load == 'pos':
        for _ in range(6): ff.pos(C)
      elif load == 'ff': ff((xm, xo))
    e._side.wait_stream(torch.cuda.current_stream()) if load == 'none' else None
    for o in outs:
      e.render_heightmap(poses, mesh, nb, out=o)
    with torch.no_grad():
      if load == 'pos':
        for _ in range(6): ff.pos(C)
      elif load == 'ff': ff((xm, xo))
    torch.cuda.synchronize()
    for o in outs:
      if not torch.equal(o.view(torch.int32), ref.view(torch.int32)):
        dm = (o.view(torch.int32) != ref.view(torch.int32)).reshape(B, -1)
        bad += int(dm.any(1).sum())
        if shown < 6:
          for ei in dm.any(1).nonzero()[:2, 0].tolist():
            px = dm[ei].nonzero()[:, 0]
            a, b_ = o[ei].reshape(-1)[px], ref[ei].reshape(-1)[px]
            rows, cols = (px // 128), (px % 128)
            print('   env', ei, 'nb', int(nb[ei]), 'pixels', int(px.numel()), 'rows', int(rows.min()), int(rows.max()), 'cols', int(cols.min()), int(cols.max()),
                  'max |dH|', float((a - b_).abs().max()), 'got', a[:4].tolist(), 'ref', b_[:4].tolist(), flush=True)
            shown += 1
  print('load', load, ':', N, 'launches x', B, 'envs; env-renders that differ from the first launch:', bad, flush=True)
