import sys, os, time, numpy as np, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def work(a):
  n, L, seed, offset = a
  from oracle.oracle import OracleEnv
  from stackrl_amd import assets
  from stackrl_amd.config import StackConfig
  pool = assets.default_pool()
  cfg = StackConfig(n_envs=n, episode_length=L, env_index_offset=offset)
  env = OracleEnv(cfg, pool, seed=seed)
  env.reset()
  AW = cfg.overhead_res - cfg.object_res + 1; r = cfg.object_res
  sub = np.zeros((L, n), np.int32); sw = np.zeros((L, n), np.int32); F = np.zeros((L, n, 10), np.float32)
  for t in range(L):
    a = env.sample()
    Hm, Om, g = env.maps()
    if offset == 0 and t == 0: print('Om', Om[0].min(), Om[0].max(), np.unique(Om[0])[:5], 'Hm', Hm[0].min(), Hm[0].max())
    for e in range(n):
      u, v = divmod(int(a[e]), AW)
      P = Hm[e, u:u + r, v:v + r]; O = Om[e]
      m = O > O.min()  # footprint guess
      if m.sum() == 0: m = np.ones_like(O, bool)
      s = P + 0.0
      d = (P - O)[m]
      S = np.where(m, P + O, -1.0)          # release surface: the rock comes to touch at the argmax (observer.py:405-413)
      k = int(S.argmax()); ci, cj = divmod(k, r)
      ii, jj = np.nonzero(m)
      gap = S.max() - S[m]
      off = np.hypot(ci - ii.mean(), cj - jj.mean())
      F[t, e] = [P[m].max(), P[m].mean(), d.max(), np.sort(d)[-max(1, d.size // 10):].mean() - d.max(), (P[m] > 0).mean(), P.max(),
                 gap.mean(), off, S.max(), gap.mean() * off]
    env.step(a)
    s = env.state()[2]
    sub[t] = s.sum(1); sw[t] = env.sweeps()
  return sub, sw, F
if __name__ == '__main__':
  L = int(sys.argv[1]); per = int(sys.argv[2])
  with mp.get_context('fork').Pool(8) as p:
    res = p.map(work, [(per, L, 1234, i * per) for i in range(8)])
  np.savez('/tmp/feat_L%d.npz' % L, sub=np.concatenate([r[0] for r in res], 1), sw=np.concatenate([r[1] for r in res], 1), F=np.concatenate([r[2] for r in res], 1))
