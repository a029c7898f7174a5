#!/usr/bin/env python3
"""Mean per-launch SQ instruction counters of one kernel from rocprofv3 --pmc passes (one directory per pass).
usage: pmc_insts.py <kernel> <dir> [<dir> ...]"""
import sys, glob, csv, os, collections
kernel = sys.argv[1]
acc = collections.defaultdict(list)
for d in sys.argv[2:]:
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
      if row['Kernel_Name'].startswith(kernel):
        acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
  v = acc[k]
  print('%-28s mean %14.1f  min %14.1f  max %14.1f  (n=%d)' % (k, sum(v) / len(v), min(v), max(v), len(v)))
if 'SQ_WAVES' in acc:
  w = sum(acc['SQ_WAVES']) / len(acc['SQ_WAVES'])
  for k in sorted(acc):
    if k.startswith('SQ_INSTS'):
      print('%-28s per wave %10.1f' % (k, sum(acc[k]) / len(acc[k]) / w))
