#!/usr/bin/env python3
"""Per-kernel duration statistics of a rocprofv3 --kernel-trace directory, BY LAUNCH SIZE.

usage: kernel_stats.py <dir> [--prefix srl_k_] > profiles/rNN_bench_kernel_stats.csv

rocprofv3's own `--stats` summary averages every launch of a kernel name; bench.py launches the env kernels at several batch
sizes (1,024 envs in the timed loop, 64 in the reward-MSE leg, 256 per handle in the free-running leg), and a mean over those
is no launch's figure (round 4's r04_bench_kernel_stats.csv said 23.0 us for a render launch the bench line timed at 33.8).
This prints rocprofv3's columns with one row per (kernel, workgroups per launch), so that the row of the headline batch can be
compared with the line's `avg_launch_us` directly."""
import collections
import csv
import glob
import math
import os
import sys

args = sys.argv[1:]
prefix = ''
if '--prefix' in args:
  i = args.index('--prefix'); prefix = args[i + 1]; args = args[:i] + args[i + 2:]
rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(args[0], '**', '*kernel_trace.csv'), recursive=True):
  for r in csv.DictReader(open(f)):
    name = r['Kernel_Name']
    if not name.startswith(prefix):
      continue
    wg = 1
    for ax in 'XYZ':
      wg *= max(1, int(r['Grid_Size_' + ax]) // max(1, int(r['Workgroup_Size_' + ax])))
    rows[(name.split('(')[0], wg)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
total = sum(sum(v) for v in rows.values()) or 1
w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(['Name', 'Workgroups', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
for (name, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
  m = sum(v) / len(v)
  sd = math.sqrt(sum((x - m) ** 2 for x in v) / len(v))
  w.writerow([name, wg, len(v), sum(v), round(m, 3), round(100.0 * sum(v) / total, 4), min(v), max(v), round(sd, 3)])
