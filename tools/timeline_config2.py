#!/usr/bin/env python3
"""Who runs when in the DQN loop (configs[2] is the sum of what the device has to do).  Reads a rocprofv3
--kernel-trace directory of `bench.py --config 2 ...` and, over the last `iters` iterations' worth of time, reports: wall,
time with any kernel running, with a settle kernel running, with a forward / update kernel running, with BOTH classes running,
with settle alone, with network kernels alone, idle; and the summed kernel time per class.

usage: timeline_config2.py <dir> [window_ms=700]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 700.0
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
  for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# the DQN leg ends the trace (leg A runs first); take the last `win` ms before the last settle kernel of 2,048-env launches
t_end = max(e for s, e, n in rows if n.startswith('srl_k_step'))
t0 = t_end - int(win * 1e6)
sel = [(max(s, t0), min(e, t_end), n) for s, e, n in rows if e > t0 and s < t_end]


def cls(n):
  if n.startswith('srl_k_step') or n.startswith('srl_k_order'):
    return 'settle'
  if n.startswith('srl_k_'):
    return 'render'
  if 'k_t' in n and ('k_tconv' in n or 'k_twrw' in n or 'k_tact' in n or 'k_thead' in n or 'k_tvalue' in n or 'k_trepack' in n or 'k_tlayout' in n or 'k_tcorr' in n or 'k_tflip' in n or 'k_tu8' in n or 'k_td_' in n):
    return 'update'
  if 'k_adam' in n or 'k_gumbel' in n or 'k_replay' in n or 'k_logit' in n:
    return 'update'
  if 'k_' in n:
    return 'forward'
  return 'other'


ev = []
for s, e, n in sel:
  c = cls(n)
  ev.append((s, 1, c)); ev.append((e, -1, c))
ev.sort()
cnt = {'settle': 0, 'render': 0, 'update': 0, 'forward': 0, 'other': 0}
acc = {}
last = t0
for t, dlt, c in ev:
  key = tuple(sorted(k for k, v in cnt.items() if v > 0))
  acc[key] = acc.get(key, 0) + (t - last)
  last = t
  cnt[c] += dlt
acc[()] = acc.get((), 0) + (t_end - last)
tot = float(t_end - t0)
print('window %.1f ms' % (tot / 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
  print('  %-40s %7.1f ms  %5.1f %%' % ('+'.join(k) or 'idle', v / 1e6, 100.0 * v / tot))
summed = {}
for s, e, n in sel:
  summed[cls(n)] = summed.get(cls(n), 0) + (e - s)
print('summed kernel time per class (overlaps counted per kernel):')
for k, v in sorted(summed.items(), key=lambda kv: -kv[1]):
  print('  %-10s %8.1f ms' % (k, v / 1e6))
# settle launches in the window
st = [(s, e) for s, e, n in sel if n.startswith('srl_k_step')]
print('settle launches: %d, mean %.1f ms' % (len(st), sum(e - s for s, e in st) / max(len(st), 1) / 1e6))
