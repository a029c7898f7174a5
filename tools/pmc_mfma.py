#!/usr/bin/env python3
"""MFMA busy cycles of the Q-net rollout forward from a rocprofv3 --pmc pass.

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> \\
      -- python3 tools/profile_qnet.py run 512 bf16
  python3 tools/pmc_mfma.py <dir>

Kernels between the two marker launches (arange of 12,345 / 23,456 elements) are the timed forwards.  SQ_VALU_MFMA_BUSY_CYCLES
counts cycles summed over the SIMDs (MI355X_MICROARCH.md); utilisation = busy cycles / (1,024 SIMDs x kernel cycles), kernel
cycles = the dispatch durations of the same run's kernel trace x 2.4 GHz (the MI355X engine clock; GRBM_GUI_ACTIVE, which
this pass also collects, is reported summed over the 8 XCDs and agrees with that within a few per cent once divided by 8)."""
import sys, glob, csv, os, collections
d = sys.argv[1]
js = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
CLOCK = 2.4e9
tr = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
dur = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9 * CLOCK for r in csv.DictReader(open(tr))}
f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# dispatches in order; find the markers
byid = collections.OrderedDict()
for r in rows:
  k = int(r['Dispatch_Id'])
  e = byid.setdefault(k, {'name': r['Kernel_Name']})
  e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
ids = sorted(byid)
marks = [i for i in ids if 'arange' in byid[i]['name']]
a, b = marks[-2], marks[-1]
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i in ids:
  if a < i < b:
    e = byid[i]
    c = agg[e['name'][:90]]
    c[0] += e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0); c[1] += dur.get(i, 0.0); c[2] += 1
    c.append(e.get('GRBM_GUI_ACTIVE', 0.0))
tm = sum(v[0] for v in agg.values()); tg = sum(v[1] for v in agg.values()); gr = sum(sum(v[3:]) for v in agg.values())
print('timed region (3 forwards of 512 samples): MFMA busy cycles %.3g summed over the SIMDs; kernel time %.2f ms = %.3g cycles at 2.4 GHz '
      '(GRBM_GUI_ACTIVE / 8 XCDs: %.3g)  ->  MFMA pipes busy %.1f %% of 1,024 SIMDs x kernel time' % (tm, tg / CLOCK * 1e3, tg, gr / 8, 100 * tm / (1024 * tg)))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
  m, g, c = v[0], v[1], v[2]
  print('  %5.1f %% of its own time  %5.1f %% of all MFMA cycles  %4d launches  %s' % (100 * m / (1024 * g) if g else 0, 100 * m / tm if tm else 0, c, n))

if js:
  import json
  json.dump({'mfma_busy_frac': tm / (1024 * tg) if tg else None, 'mfma_busy_cycles': tm, 'kernel_ms': tg / CLOCK * 1e3,
             'top_kernels': [{'kernel': n, 'mfma_busy_frac_of_own_time': (v[0] / (1024 * v[1]) if v[1] else None),
                              'share_of_mfma_cycles': v[0] / tm if tm else None, 'launches': v[2]}
                             for n, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]],
             'how': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/profile_qnet.py run 512 <dtype>: '
                    'busy cycles summed over the SIMDs / (1,024 SIMDs x kernel time x 2.4 GHz), three forwards of 512 samples',
             'commit': os.environ.get('SRL_COMMIT')}, open(js, 'w'), indent=2)
