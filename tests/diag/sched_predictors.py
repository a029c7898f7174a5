import numpy as np
from scipy.stats import spearmanr
d = np.load('/tmp/feat_L16.npz'); sub, sw, F = d['sub'].astype(float), d['sw'].astype(float), d['F']
c = 1.6 * sw + 15 * sub
L, N = c.shape
names = ['Pmax', 'Pmean', 'dmax', 'd10-dmax', 'cover', 'Pall', 'gapmean', 'off', 'Smax', 'gapoff']
for t in [0, 1, 3, 7, 11, 15]:
  print(t, 'mean %.0f max %.0f' % (c[t].mean(), c[t].max()), ' '.join('%s %.2f' % (n, spearmanr(F[t, :, k], c[t])[0]) for k, n in enumerate(names)), 'prev %.2f' % (spearmanr(c[t - 1], c[t])[0] if t else 0))
import heapq
def fcfs(c, C, order):
  h = [0.0] * C; heapq.heapify(h); end = 0.0
  for j in order:
    t = heapq.heappop(h); e = t + c[j]; end = max(end, e); heapq.heappush(h, e)
  return end
C = N // 4
tot = {}
for name, key in [('random', None), ('Pmax', 0), ('Pmean', 1), ('dmax', 2), ('Pall', 5), ('gapmean', 6), ('off', 7), ('Smax', 8), ('gapoff', 9), ('Pmax+dmax', -1), ('LPT', -2)]:
  r = []
  for t in range(L):
    if key is None: o = np.arange(N)
    elif key == -2: o = np.argsort(-c[t])
    elif key == -1: o = np.argsort(-(F[t, :, 0] + F[t, :, 2]))
    else: o = np.argsort(-F[t, :, key], kind='stable')
    r.append(fcfs(c[t], C, o))
  print(name, 'sum over episode %.1f ms' % (sum(r) / 1e3), ' '.join('%.1f' % (x / 1e3) for x in r))
