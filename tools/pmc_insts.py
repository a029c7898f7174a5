#!/usr/bin/env python3
"""Mean per-launch SQ instruction counters of one kernel from rocprofv3 --pmc passes (one directory per pass).
usage: pmc_insts.py <kernel> <dir> [<dir> ...] [--json out.json] [--wgs N]
--wgs N keeps only the launches of N workgroups (a mean over launches of different batch sizes is no launch's figure).

With --json: a summary for bench.py's line (`settle.valu_util`, `settle.wait_frac`): VALU utilisation = SQ_INSTS_VALU x 4
cycles (one 64-lane instruction occupies a 16-lane SIMD for four cycles) / (1,024 SIMDs x the launch's duration from the same
passes' kernel trace x 2.4 GHz); waiting share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (both in quad-cycles, summed over the waves).
The commit the passes were taken at comes from SRL_COMMIT (set by the refresh script from `git rev-parse` before the box is
asked for: the GPU box has no .git)."""
import sys, glob, csv, os, collections
import json
args = sys.argv[1:]
js = None
if '--json' in args:
  i = args.index('--json'); js = args[i + 1]; args = args[:i] + args[i + 2:]
WGS = None
if '--wgs' in args:
  i = args.index('--wgs'); WGS = int(args[i + 1]); args = args[:i] + args[i + 2:]
kernel = args[0]
acc = collections.defaultdict(list)
dur = []
for d in args[1:]:
  for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
      if row['Kernel_Name'].startswith(kernel):
        if WGS is not None and int(row['Grid_Size_X']) * int(row['Grid_Size_Y']) * int(row['Grid_Size_Z']) != WGS * int(row['Workgroup_Size_X']) * int(row['Workgroup_Size_Y']) * int(row['Workgroup_Size_Z']):
          continue
        dur.append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-9)
for d in args[1:]:
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
      if row['Kernel_Name'].startswith(kernel):
        if WGS is not None and int(row['Grid_Size']) != WGS * int(row['Workgroup_Size']):
          continue
        acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
  v = acc[k]
  print('%-28s mean %14.1f  min %14.1f  max %14.1f  (n=%d)' % (k, sum(v) / len(v), min(v), max(v), len(v)))
if 'SQ_WAVES' in acc:
  w = sum(acc['SQ_WAVES']) / len(acc['SQ_WAVES'])
  for k in sorted(acc):
    if k.startswith('SQ_INSTS'):
      print('%-28s per wave %10.1f' % (k, sum(acc[k]) / len(acc[k]) / w))

if js:
  mean = lambda k: (sum(acc[k]) / len(acc[k])) if acc.get(k) else None
  d = sum(dur) / len(dur) if dur else None
  valu = mean('SQ_INSTS_VALU')
  out = {'kernel': kernel, 'launches': len(dur), 'workgroups_per_launch': WGS, 'avg_launch_ms': None if d is None else 1e3 * d,
         'valu_insts_per_launch': valu,
         'valu_util': None if (valu is None or d is None) else valu * 4.0 / (1024 * d * 2.4e9),
         'wait_frac': None if (mean('SQ_WAIT_ANY') is None or not mean('SQ_WAVE_CYCLES')) else mean('SQ_WAIT_ANY') / mean('SQ_WAVE_CYCLES'),
         'lds_insts_per_launch': mean('SQ_INSTS_LDS'), 'salu_insts_per_launch': mean('SQ_INSTS_SALU'), 'waves': mean('SQ_WAVES'),
         'how': 'separate rocprofv3 --kernel-trace --pmc passes of bench.py (tools/refresh_profiles_r05.sh): valu_util = SQ_INSTS_VALU x 4 '
                '/ (1,024 SIMDs x launch duration x 2.4 GHz); wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES',
         'commit': os.environ.get('SRL_COMMIT')}
  json.dump(out, open(js, 'w'), indent=2)
