import sys, os, time, numpy as np, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def work(a):
  n, L, seed, offset = a
  from oracle.oracle import OracleEnv
  from stackrl_amd import assets
  from stackrl_amd.config import StackConfig
  pool = assets.default_pool()
  env = OracleEnv(StackConfig(n_envs=n, episode_length=L, env_index_offset=offset), pool, seed=seed)
  env.reset()
  sub = np.zeros((L, n), np.int32); sw = np.zeros((L, n), np.int32)
  for t in range(L):
    env.step(env.sample())
    s = env.state()[2]
    sub[t] = s.sum(1)
    sw[t] = env.sweeps()
    if t == 0: print('sub cols', s[:3])
  return sub, sw
if __name__ == '__main__':
  L = int(sys.argv[1]); per = int(sys.argv[2])
  t0 = time.time()
  with mp.get_context('fork').Pool(8) as p:
    res = p.map(work, [(per, L, 1234, i * per) for i in range(8)])
  sub = np.concatenate([r[0] for r in res], 1); sw = np.concatenate([r[1] for r in res], 1)
  np.savez('/tmp/dist_L%d.npz' % L, sub=sub, sw=sw)
  print('t', time.time() - t0, sub.shape, sub.mean(), sub.max(1).mean(), sw.mean(), (sw.sum()/sub.sum()))
