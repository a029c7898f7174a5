#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) for one kernel.

usage: pmc_summary.py <fetch_dir> <write_dir> <kernel> <alg_bytes_per_launch> [--wgs N] > profiles/rNN_render_pmc.json
--wgs N keeps only the launches of N workgroups (Grid_Size / Workgroup_Size of the counter rows): bench.py launches the kernel
at other batch sizes too (the 64-env reward-MSE leg, the free-running handles of 256 envs), and a mean over launches of
different sizes is not a per-launch figure of any of them (round 4's files were such means).
FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request for wide reads and is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import sys, glob, csv, json, os


def collect(d, counter, kernel, wgs=None):
  vals = []
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
      if row['Kernel_Name'].startswith(kernel) and row['Counter_Name'] == counter:
        if wgs is not None and int(row['Grid_Size']) != wgs * int(row['Workgroup_Size']):
          continue
        vals.append(float(row['Counter_Value']))
  return vals


argv = sys.argv[1:]
WGS = None
if '--wgs' in argv:
  i = argv.index('--wgs'); WGS = int(argv[i + 1]); argv = argv[:i] + argv[i + 2:]
fd, wd, kernel, alg = argv[0], argv[1], argv[2], float(argv[3])
_collect = collect
collect = lambda d, c, k: _collect(d, c, k, WGS)
f = collect(fd, 'FETCH_SIZE', kernel); w = collect(wd, 'WRITE_SIZE', kernel)
mean = lambda v: sum(v) / max(len(v), 1)
out = {
  'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --no-cpu --no-dqn --steps 18 (SRL_NO_FREE_RUN=1)',
  'kernel': kernel, 'launches': min(len(f), len(w)), 'workgroups_per_launch': WGS,
  'WRITE_SIZE_KB_mean': round(mean(w)), 'WRITE_SIZE_KB_min': round(min(w)), 'WRITE_SIZE_KB_max': round(max(w)),
  'FETCH_SIZE_KB_mean': round(mean(f)), 'FETCH_SIZE_KB_min': round(min(f)), 'FETCH_SIZE_KB_max': round(max(f)),
  'fetch_correction': 'gfx950 FETCH_SIZE counts 64 B per 128-B request for wide streaming reads (MI355X_MICROARCH.md, HBM): doubled',
  'traffic_bytes_per_launch': int((mean(w) + 2 * mean(f)) * 1024),
  'traffic_formula': '(WRITE_SIZE + 2 * FETCH_SIZE) * 1024',
  'algorithmic_bytes_per_launch': int(alg),
  'traffic_over_algorithmic': round((mean(w) + 2 * mean(f)) * 1024 / alg, 4),
  'commit': os.environ.get('SRL_COMMIT'),
}
print(json.dumps(out, indent=2))
