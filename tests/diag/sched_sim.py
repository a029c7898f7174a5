import numpy as np, heapq
d = np.load('/tmp/dist_L16.npz'); sub = d['sub'].astype(float); sw = d['sw'].astype(float)
L, N = sub.shape
def cost(c_sw, c_sub): return c_sw * sw + c_sub * sub      # us
def fcfs(c, C, order=None):
  idx = np.arange(len(c)) if order is None else order
  h = [0.0] * C; heapq.heapify(h)
  end = 0.0
  for j in idx:
    t = heapq.heappop(h); e = t + c[j]; end = max(end, e); heapq.heappush(h, e)
  return end
def ps(c, C):
  # processor sharing bound: T = int_0^max max(1, N(x)/C) dx
  x = np.sort(c); n = len(x); T = 0.0; prev = 0.0
  for k, v in enumerate(x):
    alive = n - k
    T += (v - prev) * max(1.0, alive / C); prev = v
  return T
for c_sw, c_sub in [(1.6, 15.0), (1.9, 10.0), (2.2, 8.0)]:
  c = cost(c_sw, c_sub)
  m1024 = np.mean([c[t, :1024].max() for t in range(L)]) / 1e3
  f = np.mean([fcfs(c[t], 1024) for t in range(L)]) / 1e3
  lpt = np.mean([fcfs(c[t], 1024, np.argsort(-c[t])) for t in range(L)]) / 1e3
  p = np.mean([ps(c[t], 1024) for t in range(L)]) / 1e3
  lb = np.mean([max(c[t].max(), c[t].sum() / 1024) for t in range(L)]) / 1e3
  print('c', c_sw, c_sub, 'max@1024 %.1f  fcfs@4096 %.1f  lpt %.1f  ps %.1f  lower %.1f  work/1024 %.1f' % (m1024, f, lpt, p, lb, np.mean(c.sum(1)) / 1024e3))
