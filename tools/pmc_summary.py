#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) for one kernel.

usage: pmc_summary.py <fetch_dir> <write_dir> <kernel> <alg_bytes_per_launch> > profiles/rNN_render_pmc.json
FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request for wide reads and is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import sys, glob, csv, json, os


def collect(d, counter, kernel):
  vals = []
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
      if row['Kernel_Name'].startswith(kernel) and row['Counter_Name'] == counter:
        vals.append(float(row['Counter_Value']))
  return vals


fd, wd, kernel, alg = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
f = collect(fd, 'FETCH_SIZE', kernel); w = collect(wd, 'WRITE_SIZE', kernel)
mean = lambda v: sum(v) / max(len(v), 1)
out = {
  'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --no-cpu --steps 18',
  'kernel': kernel, 'launches': min(len(f), len(w)),
  'WRITE_SIZE_KB_mean': round(mean(w)), 'WRITE_SIZE_KB_min': round(min(w)), 'WRITE_SIZE_KB_max': round(max(w)),
  'FETCH_SIZE_KB_mean': round(mean(f)), 'FETCH_SIZE_KB_min': round(min(f)), 'FETCH_SIZE_KB_max': round(max(f)),
  'fetch_correction': 'gfx950 FETCH_SIZE counts 64 B per 128-B request for wide streaming reads (MI355X_MICROARCH.md, HBM): doubled',
  'traffic_bytes_per_launch': int((mean(w) + 2 * mean(f)) * 1024),
  'traffic_formula': '(WRITE_SIZE + 2 * FETCH_SIZE) * 1024',
  'algorithmic_bytes_per_launch': int(alg),
  'commit': os.environ.get('SRL_COMMIT'),
}
print(json.dumps(out, indent=2))
