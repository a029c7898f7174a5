#!/usr/bin/env python3
"""bench.py — env steps/sec of the batched Stack-v0 hot path (BASELINE.json metric) + the DQN rollout/update leg.

  python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4]

Launch.  `--gpus N` with N > 1 starts N fresh child processes itself (one per GPU, rendezvous on 127.0.0.1; the
parent never touches the GPU and never execs); under `python -m torch.distributed.run` (RANK / WORLD_SIZE already in
the environment) the process is a rank and starts nothing.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json `configs`):
  leg A  configs[1] — Stack-v0, 1,024 vectorised envs per GPU, 8 rocks, random policy on device, physics + render
         kernels only.  A "step" is one vectorised `env.step` (settle kernel + render kernel).  This is the
         configuration the metric's target is quoted on (>= 50 k env steps/s on one MI355X at >= 40 % of the HBM roofline
         for the render kernel) and it is `value` by default: placements of all ranks / max-over-ranks wall time of the
         K timed steps (auto-reset calls, env.py:235-236, are stepped and timed but are not placements).  Weak scaling:
         every rank owns its own 1,024 envs; envs are independent (utils.py:424-448), no data-path collective.
  leg B  the DQN configs — rollout (Q-net forward on every env) + one minibatch-32 update per iteration
         (training.py:338-380; the rank's envs as two handles served as their steps finish, the update's gradient half
         beside the collect step: `env_groups`, `update_early_gradient` in the line), reported under "dqn": configs[2] (4,096 envs x 16 rocks) at N = 1, configs[3]'s per-GPU
         shard (2,048 x 16) at N = 2 / 4, configs[4]'s (2,048 x 32 rocks, 64 x 64 maps) at N = 8; the one collective is
         the RCCL all-reduce of the flat gradient bucket.  Rollout in fp32-class precision (bf16x3 products) and, as a second
         labelled entry, bf16.  `--config 2|3|4` makes that leg the headline `value` instead.

Extra objects on the JSON line: `roofline` (render kernel K2, HBM-bound: algorithmic bytes / HIP-event time measured in
the timed region), `settle` (K1 launch statistics), `cpu_baseline` (the CPU oracle timed on this
host's cores on a bounded sample of leg A's workload), `reward_mse_vs_cpu`, `dqn` (leg B incl. its own `roofline` with
bound "mfma").
"""
import argparse
import json
import os
import socket
import subprocess
import threading
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Leg B keeps five HIP streams busy per rank (current, two env groups, the update's gradient half, RCCL's own); HIP
# multiplexes streams over GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams on one queue run their kernels one
# after the other — a settle launch of tens of milliseconds in front of the forward.  A documented ROCm knob, read when the
# runtime initialises: set here, before anything touches the GPU, unless the caller chose a value.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E (MI355X_MICROARCH.md)
# dense MFMA peaks, same guide.  The fp32-class rollout computes every product as three bf16 MFMAs (hi hi + hi lo + lo hi,
# csrc/conv_mfma.hip, conv_gemm.hip, xcorr_mfma.hip): its ceiling is a third of the bf16 peak, not the fp32 MFMA peak (157.3)
PEAK_TFLOPS = {'bf16x3': 2500.0 / 3.0, 'bf16': 2500.0}


def parse_args(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=54)
  ap.add_argument('--warmup', type=int, default=9)
  ap.add_argument('--config', type=int, default=1, choices=[1, 2, 3, 4],
                  help='BASELINE.json configs[i] whose throughput is the headline `value` (1 = env only)')
  ap.add_argument('--envs', type=int, default=None, help='leg A: envs per GPU (default 1024)')
  ap.add_argument('--rocks', type=int, default=None, help='leg A: episode_length (default 8)')
  ap.add_argument('--res', type=int, default=128, choices=[64, 128], help='leg A: overhead map side')
  ap.add_argument('--solver', default='pybullet', help="solver definition: 'pybullet' (default) or 'bullet10'")
  ap.add_argument('--seed', type=int, default=11)
  ap.add_argument('--launch-order', default='auto', choices=['auto', 'index', 'ordered'],
                  help='settle workgroups take the envs in index order, highest release first, or by batch size (srl_set_launch_order)')
  ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline / reward-MSE legs')
  ap.add_argument('--no-dqn', action='store_true', help='skip leg B')
  ap.add_argument('--dqn-iters', type=int, default=None, help='leg B: timed iterations (default: one whole episode, L + 1 calls)')
  ap.add_argument('--dqn-warmup', type=int, default=6)
  ap.add_argument('--dqn-envs', type=int, default=None, help='leg B: envs per GPU (default by --gpus, see above)')
  ap.add_argument('--dqn-rocks', type=int, default=None)
  ap.add_argument('--dqn-res', type=int, default=None, choices=[64, 128])
  ap.add_argument('--dqn-slots', type=int, default=16, help='replay capacity in transitions per env')
  ap.add_argument('--dqn-groups', type=int, default=None,
                  help='leg B: the rank\'s envs as this many handles that step as their actions arrive (PipelinedVecStackEnv); '
                       'default 2, 1 = one handle')
  ap.add_argument('--rollout', default='both', choices=['bf16x3', 'f32', 'bf16', 'both'],
                  help="rollout precision: 'bf16x3' = fp32-class (every product as three bf16 MFMAs; 'f32' is an alias), 'bf16'")
  ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='nccl = RCCL; gloo for rehearsals')
  ap.add_argument('--same-device', action='store_true', help='rehearsal: every rank uses device 0 (needs --backend gloo)')
  ap.add_argument('--leg-b-timeout', type=float, default=900.0,
                  help='N > 1: seconds leg B (DQN) may take after leg A before the launcher ends the ranks and prints leg A')
  ap.add_argument('--launch-only', action='store_true',
                  help='rehearsal without a GPU: spawn, rendezvous, barriers and aggregation only; value is null')
  return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher
PROVISIONAL = '#provisional '


def free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def launch(args, argv):
  """Start one fresh child per rank (plain subprocesses of this same script: nothing here has touched the GPU, and the
  children are started, not exec'd into) and wait for them; rank 0's stdout is this process's stdout."""
  port = free_port()
  procs = []
  for r in range(args.gpus):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if args.same_device else r), WORLD_SIZE=str(args.gpus),
               LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               SRL_BENCH_CHILD='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
  got = {'final': None, 'provisional': None}

  def reader():
    for raw in procs[0].stdout:
      text = raw.decode(errors='replace').rstrip('\n')
      if text.startswith(PROVISIONAL):
        got['provisional'] = text[len(PROVISIONAL):]
        got['since'] = time.time()
      elif text.strip():
        got['final'] = text

  th = threading.Thread(target=reader, daemon=True)
  th.start()
  rc = 0
  deadline = None
  while any(p.poll() is None for p in procs):
    time.sleep(0.2)
    failed = [p for p in procs if p.poll() not in (None, 0)]
    if failed and deadline is None:
      deadline = time.time() + 20.0          # a rank died: give the others a moment, then end exactly those we started
    if deadline is None and got.get('since') and time.time() - got['since'] > args.leg_b_timeout:
      print('bench.py: leg B exceeded --leg-b-timeout ({} s): ending the ranks'.format(args.leg_b_timeout), file=sys.stderr)
      deadline = time.time()
    if deadline is not None and time.time() > deadline:
      for p in procs:
        if p.poll() is None:
          p.kill()
  th.join(timeout=5.0)
  for p in procs:
    rc = rc or (p.returncode or 0)
  if got['final'] is not None:
    print(got['final'], flush=True)
  elif got['provisional'] is not None:
    # leg A finished on every rank and was aggregated; a rank ended in leg B: the headline stands, the line says so
    print(got['provisional'], flush=True)
    print('bench.py: a rank ended during leg B (exit codes {}); the line above carries leg A only'.format(
      [p.returncode for p in procs]), file=sys.stderr)
    return rc or 1        # the provisional line is printed, but a lost rank is not a success
  return rc


# ----------------------------------------------------------------------------------------------- helpers
def alg_bytes_per_env(res, r, nb):
  """H f32 + obs u8 (H,W,2) + O f32 + obs u8 (h,w,1) + 1,404 B per placed rock."""
  return 6 * res * res + 5 * r * r + 1404 * nb


def _cpu_worker(a):
  n, L, seed, offset, episodes, kw = a
  from oracle.oracle import OracleEnv
  from stackrl_amd import assets
  from stackrl_amd.config import StackConfig
  pool = assets.default_pool()
  env = OracleEnv(StackConfig(n_envs=n, episode_length=L, env_index_offset=offset, **kw), pool, seed=seed)
  env.reset()
  t0 = time.perf_counter()
  for _ in range(episodes * L):
    env.step(env.sample())
  return n * L * episodes, time.perf_counter() - t0


def host_cores():
  """(physical cores this process may run on, logical CPUs it may run on, cgroup CPU quota in cores or None).  Physical cores =
  distinct (physical id, core id) pairs of /proc/cpuinfo among the CPUs of the affinity mask; the quota is cpu.max (cgroup v2)
  or cfs_quota / cfs_period (v1) when the container has one."""
  try:
    allowed = sorted(os.sched_getaffinity(0))
  except AttributeError:
    allowed = list(range(os.cpu_count() or 1))
  phys, cur = set(), {}
  try:
    with open('/proc/cpuinfo') as f:
      for line in f:
        if ':' in line:
          k, v = [t.strip() for t in line.split(':', 1)]
          cur[k] = v
        elif not line.strip() and cur:
          if int(cur.get('processor', -1)) in allowed and 'core id' in cur:
            phys.add((cur.get('physical id', '0'), cur['core id']))
          cur = {}
    if cur and int(cur.get('processor', -1)) in allowed and 'core id' in cur:
      phys.add((cur.get('physical id', '0'), cur['core id']))
  except (OSError, ValueError):
    pass
  quota = None
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      q, per = f.read().split()[:2]
      if q != 'max':
        quota = float(q) / float(per)
  except (OSError, ValueError):
    try:
      with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:
        q, per = float(f.read()), float(g.read())
        if q > 0:
          quota = q / per
    except (OSError, ValueError):
      pass
  return (len(phys) or len(allowed)), len(allowed), quota


def _cpu_pool_run(cores, budget_envs, L, seed, episodes, kw):
  import multiprocessing as mp
  ctx = mp.get_context('fork')
  t0 = time.perf_counter()
  with ctx.Pool(cores) as pool:
    res = pool.map(_cpu_worker, [(budget_envs, L, seed, i * budget_envs, episodes, kw) for i in range(cores)], chunksize=1)
  wall = time.perf_counter() - t0
  return sum(r[0] for r in res), max(r[1] for r in res), wall


def cpu_baseline(L, seed, kw, budget_envs=48, episodes=20):
  """The oracle (oracle/srl_oracle.c, built -O3 at the x86-64-v3 level, oracle/Makefile) on this host's cores, forked before
  this process touches the GPU.  `value` / `cores`: one process per PHYSICAL core the process may use (the reference's own
  default is every core, envs/utils.py:342 `n_parallel or mp.cpu_count()`; BASELINE.md section 3: "1 thread and all physical
  cores"), each stepping `budget_envs` envs through `episodes` episodes of the same workload (about 10 s of work per core);
  the 16-process figure of rounds 1 - 4 and the one-thread figure are kept beside it."""
  phys, logical, quota = host_cores()
  # one process per physical core the container may actually use: under a cgroup CPU quota (the GPU boxes of this pool give a
  # one-GPU container cpu.max = 16 cores of the host's 128) more processes only share the quota's cores — measured on such a
  # box: 128 processes 13.0 k placements/s against 16.8 k with 16 (profiles/r05_cpu_quota_probe.txt)
  cores = phys if quota is None else max(1, min(phys, int(quota)))
  if os.environ.get('SRL_CPU_CORES'):                  # experiments / small boxes
    cores = max(1, int(os.environ['SRL_CPU_CORES']))
  n1, t1 = _cpu_worker((budget_envs, L, seed, 0, max(1, episodes // 10), kw))     # one-thread figure first
  placed, busy, wall = _cpu_pool_run(cores, budget_envs, L, seed, episodes, kw)
  sixteen = None
  if cores > 16:
    p16, b16, w16 = _cpu_pool_run(16, budget_envs, L, seed, episodes, kw)
    sixteen = {'value': p16 / b16, 'cores': 16, 'wall_s': round(w16, 1)}
  model = None
  try:
    with open('/proc/cpuinfo') as f:
      model = next((l.split(':', 1)[1].strip() for l in f if l.startswith('model name')), None)
  except OSError:
    pass
  return {
    'value': placed / busy, 'unit': 'env_steps/s', 'cores': cores, 'kind': 'port', 'nproc': os.cpu_count(), 'cpu_model': model,
    'physical_cores': phys, 'logical_cpus_allowed': logical, 'cgroup_cpu_quota': quota,
    # placements / the slowest process's busy time (every process does the same amount of work); by the wall clock of the
    # whole pool, process start included, it is value_by_wall
    'value_by_wall': placed / wall,
    'cores_note': ('one process per physical core' if quota is None or quota >= phys else
                   'the container\'s CPU quota (cgroup cpu.max) is {:g} cores of the host\'s {} physical cores: one process per usable core'.format(quota, phys)),
    'sample': '{} procs (one per usable physical core) x {} envs x {} episode(s) of {} placements (oracle/srl_oracle.c -O3 x86-64-v3, same pool, '
              'RNG keys and solver definition); wall incl. process start {:.1f}s'.format(cores, budget_envs, episodes, L, wall),
    'single_thread_value': n1 / t1,
    'sixteen_process_value': sixteen,
  }


def build_info():
  from stackrl_amd import build as _b
  i = _b.info(_b.LIB) or {}
  return {'variant': i.get('variant'), 'source_hash': i.get('hash')}


def aggregate(dt, placed, world, device):
  """Whole-job numbers from per-rank ones: MAX of the elapsed time over ranks, SUM of the placements.  These two tiny
  all-reduces (and the barriers) are the only collectives of the env path: envs are independent (utils.py:424-448)."""
  import torch
  import torch.distributed as dist
  t = torch.tensor([dt], dtype=torch.float64, device=device)
  tot = torch.tensor([float(placed)], dtype=torch.float64, device=device)
  if world > 1 or (dist.is_available() and dist.is_initialized()):
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
  return float(t.item()), float(tot.item())


def _latest_profile(pattern):
  """(relative path, parsed JSON) of the newest profiles/rNN_<pattern>, or (None, None).  Counter collection cannot run
  inside the timed process: these files come from separate `rocprofv3 --pmc` passes (tools/refresh_profiles_r03.sh),
  each records the commit it was taken at (`commit`, written by the tool that made it — this process starts no child
  process: it has initialised the GPU by the time the line is assembled, and the GPU box has no .git anyway)."""
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_' + pattern)))
  if not files:
    return None, None
  try:
    with open(files[-1]) as f:
      return os.path.relpath(files[-1], ROOT), json.load(f)
  except Exception:   # noqa: BLE001
    return None, None


def traffic_record(B, L, res, alg_bytes_per_launch=None):
  """HBM bytes per render launch from the PMC counters (profiles/rNN_render_pmc.json, tools/pmc_summary.py, launches of B
  workgroups only).  A counter figure outside [0.8, 1.5] x the algorithmic bytes is not printed as traffic: every byte of H and
  of the observation is compared in the parity tests, so less than the algorithmic stores is a summarising error (round 4: a
  mean over launches of two batch sizes), and the note says so."""
  if not (B == 1024 and L == 8 and res == 128):
    return None, None
  rel, d = _latest_profile('render_pmc.json')
  if d is None:
    return None, None
  src = {'file': rel, 'commit': d.get('commit'), 'workgroups_per_launch': d.get('workgroups_per_launch'),
         'how': 'separate rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE)'}
  t = d.get('traffic_bytes_per_launch')
  alg = alg_bytes_per_launch or d.get('algorithmic_bytes_per_launch')
  if t is None or d.get('workgroups_per_launch') not in (None, B) or (alg and not (0.8 <= t / alg <= 1.5)):
    src['note'] = 'counter figure {} B per launch rejected (not of {}-workgroup launches, or outside [0.8, 1.5] x the algorithmic bytes)'.format(t, B)
    return None, src
  src['traffic_over_algorithmic'] = None if not alg else round(t / alg, 4)
  return t, src


def settle_counters(B, L, res):
  """VALU utilisation and waiting share of srl_k_step from its PMC passes (profiles/rNN_settle_pmc.json, tools/pmc_insts.py).
  Only a summary over launches of this batch (two waves per env at 8 rocks) is printed; a mean over launches of several batch
  sizes (round 4's last file: 914.8 waves per launch) is rejected with a note."""
  if not (B == 1024 and L == 8 and res == 128):
    return {}
  rel, d = _latest_profile('settle_pmc.json')
  if d is None:
    return {}
  src = {'file': rel, 'commit': d.get('commit'), 'how': d.get('how'), 'workgroups_per_launch': d.get('workgroups_per_launch')}
  waves = d.get('waves')
  if waves is not None and abs(waves - 2 * B) > 0.5:
    src['note'] = 'rejected: the summary averages launches of several batch sizes ({} waves per launch, this batch has {})'.format(waves, 2 * B)
    return {'valu_util': None, 'wait_frac': None, 'counters_source': src}
  return {'valu_util': d.get('valu_util'), 'wait_frac': d.get('wait_frac'), 'counters_source': src}


def mfma_counters(dtype):
  """MFMA-pipe busy fraction of the rollout forward from its PMC pass (profiles/rNN_qnet_mfma_pmc_<dtype>.json, tools/pmc_mfma.py)."""
  rel, d = _latest_profile('qnet_mfma_pmc_{}.json'.format({'bf16x3': 'fp32'}.get(dtype, dtype)))   # tools/profile_qnet.py's names
  if d is None:
    return {}
  return {'mfma_busy_frac': d.get('mfma_busy_frac'),
          'mfma_busy_source': {'file': rel, 'commit': d.get('commit'), 'how': d.get('how')}}


# ----------------------------------------------------------------------------------------------- leg A
def env_leg(args, rank, world, pool, barrier, solver_kw):
  import numpy as np
  import torch
  from stackrl_amd import env as envs
  B, L = args.envs or 1024, args.rocks or 8
  kw = dict(solver_kw)
  if args.res == 64:
    kw['resolution_factor'] = 4
  if args.launch_order != 'auto':
    kw['launch_order'] = args.launch_order == 'ordered'
  env = envs.VecStackEnv(n_parallel=B, seed=args.seed, pool=pool, block=False, episode_length=L,
                         env_index_offset=rank * B, **kw)
  phase = {'k': 0}   # lock-step bookkeeping: calls since reset(): 1..L placements, L+1 auto-reset

  def do_step():
    out = env.step(env.sample())
    phase['k'] += 1
    if phase['k'] == L + 1:
      phase['k'] = 0
      return out, 0, 0
    return out, B, phase['k']

  env.reset()()
  for _ in range(args.warmup):
    do_step()[0]
  env._lib.srl_sync_status(env._h, env._stream())
  env.kernel_times()
  env.set_profiling(True)
  barrier()
  t0 = time.perf_counter()
  placed, alg, rocks = 0, 0, 0
  res, r = env.config.overhead_res, env.config.object_res
  last = None
  for _ in range(args.steps):
    last, p, nb = do_step()
    placed += p
    rocks += nb
    alg += B * alg_bytes_per_env(res, r, nb)
  barrier()
  dt = time.perf_counter() - t0
  last()     # raises if any env diverged / action invalid
  ms, nl = env.kernel_times()
  env.set_profiling(False)
  # the same loop over a window of 54 calls (six whole episodes: every fill level of the scene weighs the same), reported
  # beside the contract's K-step figure as value_long
  long_steps, long_value = 54, None
  if args.steps != long_steps and world == 1:
    while phase['k'] != 0:
      do_step()[0]()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    lp = 0
    for _ in range(long_steps):
      last, p, _nb = do_step()
      lp += p
    last()
    torch.cuda.synchronize()
    long_value = lp / (time.perf_counter() - t1)
  # The same 1,024 envs held as G handles that are NOT kept in lock-step: with a policy that needs no observation (the random
  # policy of this leg) a group's next step waits for nothing but its own previous one, so a group's launch lasts as long as
  # the slowest of ITS envs, not of all of them.  Every handle on a stream of its own through the C-ABI (sample, step; output
  # buffers reused); env i keeps its seed, so its trajectory is the one it has in the lock-step batch.  Reported beside the
  # contract's figure, never as `value`: the reference's `ParallelEnv.step` waits for all its workers.
  free_run = None
  if world == 1 and not os.environ.get('SRL_NO_FREE_RUN'):
    import ctypes
    G = int(os.environ.get('SRL_FREE_RUN_GROUPS', 4))     # (2: +12 %, 4: +25 %, 8: -29 %, 32: -61 %: more streams than hardware queues queue behind one another)
    if B % G == 0:
      gs = [envs.VecStackEnv(n_parallel=B // G, seed=args.seed, pool=pool, block=False, episode_length=L,
                             env_index_offset=rank * B + k * (B // G), concurrent_envs=B, **kw) for k in range(G)]
      sts = [torch.cuda.Stream() for _ in range(G)]
      bufs = []
      for g_ in gs:
        n_ = B // G
        bufs.append((torch.empty(n_, dtype=torch.int64, device='cuda'), torch.empty((n_, res, res, 2), dtype=torch.uint8, device='cuda'),
                     torch.empty((n_, r, r, 1), dtype=torch.uint8, device='cuda'), torch.empty(n_, dtype=torch.float32, device='cuda'),
                     torch.empty(n_, dtype=torch.uint8, device='cuda')))
        g_.reset()()
      torch.cuda.synchronize()
      VP = ctypes.c_void_p

      def run(calls):
        for _ in range(calls):
          for g_, st_, (a_, om_, oo_, rw_, dn_) in zip(gs, sts, bufs):
            sp = VP(st_.cuda_stream)
            assert g_._lib.srl_sample(g_._h, VP(a_.data_ptr()), sp) == 0
            assert g_._lib.srl_step(g_._h, VP(a_.data_ptr()), VP(om_.data_ptr()), VP(oo_.data_ptr()), VP(rw_.data_ptr()), VP(dn_.data_ptr()), sp) == 0
      run(L + 1)                                   # one whole episode cycle: warm-up, and every handle back at an episode start
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      cycles = 6
      run(cycles * (L + 1))
      torch.cuda.synchronize()
      free_run = {'groups': G, 'value': cycles * L * B / (time.perf_counter() - t2), 'steps': cycles * (L + 1), 'unit': 'env_steps/s',
                  'note': 'the same envs as G handles on G streams, not in lock-step (random policy: no observation is waited for); '
                          'placements per second over six whole episodes, comparable with value_long'}
      for g_, st_ in zip(gs, sts):
        envs._check(g_._lib.srl_sync_status(g_._h, VP(st_.cuda_stream)))     # raises if an env diverged / an action was invalid
        g_.close()
  # statistics of one further (untimed) episode: what the stop criterion and the residual threshold asked of the kernel
  subs, sweeps = [], []
  while phase['k'] != 0:
    do_step()[0]()
  for _ in range(L):
    do_step()[0]()
    subs.append(env.state()[2].sum(1))
    sweeps.append(env.sweeps())
  do_step()[0]()
  sub, sw = np.stack(subs), np.stack(sweeps)
  out = dict(B=B, L=L, res=res, r=r, dt=dt, placed=placed, alg=alg, rocks=rocks, ms=ms, nl=nl, sub=sub, sw=sw, config=env.config,
             long_value=long_value, long_steps=long_steps, free_run=free_run)
  env.close()
  torch.cuda.synchronize()
  return out


def reward_mse(args, pool, L, solver_kw):
  """Reward parity on identical seeds: 64 envs, one episode, HIP vs the CPU oracle."""
  import numpy as np
  from oracle.oracle import OracleEnv
  from stackrl_amd import env as envs
  from stackrl_amd.config import StackConfig
  g = envs.VecStackEnv(n_parallel=64, seed=args.seed + 1, pool=pool, block=True, episode_length=L, **solver_kw)
  o = OracleEnv(StackConfig(n_envs=64, episode_length=L, **solver_kw), pool, seed=args.seed + 1)
  g.reset(); o.reset()
  se, cnt, ok = 0.0, 0, True
  for _ in range(L):
    a = g.sample()
    (gm, go), gr, gd = g.step(a)
    (om, oo), orr, od = o.step(a.cpu().numpy())
    se += float(((gr.cpu().numpy().astype(np.float64) - orr) ** 2).sum()); cnt += 64
    ok &= bool(np.array_equal(gm.cpu().numpy(), om) and np.array_equal(gd.cpu().numpy(), od))
  g.close()
  return {'value': se / cnt, 'envs': 64, 'steps': L, 'obs_and_done_bit_exact': ok}


# ----------------------------------------------------------------------------------------------- leg B
def dqn_shape(args):
  """Per-GPU shard of the BASELINE DQN configs: configs[2] at N = 1, configs[3] at N = 2 / 4, configs[4] at N = 8."""
  if args.config in (2, 3, 4):
    B, L, res = {2: (4096, 16, 128), 3: (2048, 16, 128), 4: (2048, 32, 64)}[args.config]
    name = 'configs[{}]'.format(args.config)
  elif args.gpus >= 8:
    B, L, res, name = 2048, 32, 64, 'configs[4]'
  elif args.gpus >= 2:
    B, L, res, name = 2048, 16, 128, 'configs[3]'
  else:
    B, L, res, name = 4096, 16, 128, 'configs[2]'
  return args.dqn_envs or B, args.dqn_rocks or L, args.dqn_res or res, name


_RCCL = {}


def rccl_group(args, world):
  """The process group of leg B's gradient all-reduce: RCCL (backend 'nccl') over all ranks, one per GPU; None = the
  default (gloo) group for the one-device rehearsals (--backend gloo); at N = 1 there is no group at all."""
  import torch.distributed as dist
  if args.backend != 'nccl' or not (dist.is_available() and dist.is_initialized()):
    return None
  if 'pg' not in _RCCL:
    import torch
    _RCCL['pg'] = dist.new_group(backend='nccl')
    # the communicator is built by the first collective: here, not inside the timed region (and from here on RCCL's
    # watchdog thread is alive while the update is captured into hipGraphs)
    dist.all_reduce(torch.zeros(1, device='cuda'), group=_RCCL['pg'])
    torch.cuda.synchronize()
  return _RCCL['pg']


def dqn_leg(args, rank, world, pool, barrier, solver_kw, dtype):
  """`Training.run` (training.py:338-380) for --dqn-iters iterations after --dqn-warmup: per iteration one rollout forward
  over the rank's envs (`agent.collect`), one non-blocking vectorised env step on a side stream, one minibatch-32 update
  (`agent.train`, with the gradient all-reduce when world > 1)."""
  import torch
  import torch.distributed as dist
  from stackrl_amd import env as envs, nets, qops
  from stackrl_amd.dqn import DQN, PolynomialDecay
  from stackrl_amd.training import Trainer
  if os.environ.get('SRL_BENCH_FAIL_LEG_B') == str(rank):     # test hook: this rank is lost in leg B
    raise RuntimeError('injected leg-B failure on rank {}'.format(rank))
  B, L, res, name = dqn_shape(args)
  if os.environ.get('SRL_FWD_CU_MASK'):        # experiment hook: the current stream (forward, update) on a subset of the CUs
    import ctypes                              # eight 32-bit hex words, comma separated, bit i = CU i
    words = [int(w, 16) for w in os.environ['SRL_FWD_CU_MASK'].split(',')]
    hip = ctypes.CDLL('libamdhip64.so')
    hs = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(hs), len(words), (ctypes.c_uint32 * len(words))(*words))
    if rc:
      raise RuntimeError('hipExtStreamCreateWithCUMask: {}'.format(rc))
    torch.cuda.set_stream(torch.cuda.ExternalStream(hs.value))
  # as the headline (--config 2|3|4) the leg follows the contract's K timed / W warm-up steps; the three eager updates
  # and the graph capture of the update (DQN._GRAPH_WARMUP) then run before those, as part of the setup
  # (default: one whole episode of L placements + the auto-reset call, so that every fill level of the scene is in the window)
  iters, warm, pre = (args.steps, args.warmup, 5) if args.config != 1 else (args.dqn_iters or L + 1, args.dqn_warmup, 0)
  kw = dict(solver_kw)
  if res == 64:
    kw['resolution_factor'] = 4
  if args.launch_order != 'auto':
    kw['launch_order'] = args.launch_order == 'ordered'
  if os.environ.get('SRL_ENV_STREAM_PRIORITY'):        # experiment hook: HIP priority of the env groups' streams (-1 = high)
    kw['stream_priority'] = int(os.environ['SRL_ENV_STREAM_PRIORITY'])
  groups = args.dqn_groups if args.dqn_groups else 2
  if groups > 1:
    kw['groups'] = groups
  env = envs.make('Stack-v0', n_parallel=B, seed=args.seed, pool=pool, episode_length=L, side_stream=True,
                  env_index_offset=rank * B, **kw)
  net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
  agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=32,
              replay_memory_size=B * args.dqn_slots, discount_factor=.966667, collect_batch_size=B,
              exploration=PolynomialDecay(1.0, 400000, .1), prioritization=0.6,
              priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7 + rank,
              policy_op=qops.FusedPolicy(chunk=int(os.environ.get('SRL_POLICY_CHUNK', 2048)), autocast=torch.bfloat16 if dtype == 'bf16' else None, fast=True),
              xcorr='bf16x3', graphs=True, process_group=rccl_group(args, world), prefetch=3,   # config.gin:55-112 (prefetch :104)
              early_gradient=not os.environ.get('SRL_NO_EARLY_GRADIENT'))
  tr = Trainer(env, agent)
  tr.initialize(num_steps=4)
  if pre:
    tr.run(pre)
  ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(iters)]

  class PolicyTimer(object):
    """HIP-event pairs around every policy evaluation of the timed region (one per iteration, or one per group)"""
    def __init__(self):
      self.on, self.pairs = False, []
    def start(self):
      if self.on:
        self.pairs.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
        self.pairs[-1][0].record()
    def stop(self):
      if self.on:
        self.pairs[-1][1].record()

  # the policy evaluation of the whole batch ALONE on the device (nothing else in flight): the kernel-level figure beside the
  # in-loop one, where the forward shares the CUs with the other group's settle kernel and the update's gradient half
  step = env.reset()
  step = step() if callable(step) else step
  getattr(env, 'drain', lambda: None)()
  torch.cuda.synchronize()
  draws = agent.policy_draws(B)
  fa, fb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  with torch.no_grad():
    agent._policy(step[0], True, False, draws)
    torch.cuda.synchronize()
    fa.record()
    for _ in range(3):
      agent._policy(step[0], True, False, draws)
    fb.record()
  torch.cuda.synchronize()
  fwd_alone_ms = fa.elapsed_time(fb) / 3
  agent.policy_timer = ptimer = PolicyTimer()
  step = env.reset()
  agent.acknowledge_reset()
  t0 = None
  placed_calls = 0        # timed calls that are placements: call c after reset() is the auto-reset when c % (L + 1) == L (env.py:235-236)
  for it in range(warm + iters):
    if it >= warm and it % (L + 1) != L:
      placed_calls += 1
    if it == warm:
      step = step() if callable(step) else step
      getattr(env, 'drain', lambda: None)()
      barrier()
      t0 = time.perf_counter()
    k = it - warm
    ptimer.on = k >= 0
    agent.train_begin()                    # the gradient half of the update, on its own stream beside the collect step (early_gradient)
    step = tr.collect_step(env, step)      # policy forward(s), replay add, env step(s) on the side stream(s): they overlap the update below
    if k >= 0:
      ev[k][0].record()
    agent.train()
    if k >= 0:
      ev[k][1].record()
  tr._drain(env, step)
  barrier()
  dt = time.perf_counter() - t0
  agent.policy_timer = None
  fwd_ms = sum(a.elapsed_time(b) for a, b in ptimer.pairs) / len(ev)
  upd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / len(ev)
  # all-reduce of the gradient bucket, timed on its own after the loop (inside the update it hides in upd_ms)
  ar_ms = None
  if dist.is_available() and dist.is_initialized():     # (world 1 under SRL_BENCH_FORCE_DIST=1: exercises the RCCL call itself)
    g = agent._flat_grad
    for _ in range(3):
      dist.all_reduce(g, group=agent._pg)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
      dist.all_reduce(g, group=agent._pg)
    b.record()
    torch.cuda.synchronize()
    ar_ms = a.elapsed_time(b) / 10
  macs = sum(nets.forward_macs(H=res, h=res // 4).values())
  dt_max, steps_all = aggregate(dt, B * placed_calls, world, 'cpu')
  flops_fwd = 2.0 * macs * B
  out = {
    'workload': 'Stack-v0, {} envs per GPU x {} rocks, {}x{} maps, DQN rollout + minibatch-32 update per iteration '
                '(BASELINE {} per-GPU shard)'.format(B, L, res, res, name),
    'rollout_dtype': dtype,
    'rollout_dtype_note': ('fp32-class: every product as three bf16 MFMAs (hi*hi + hi*lo + lo*hi), ~16 mantissa bits, held to 2e-5 ... 3e-5 of fp64 per layer in tests — tighter than bf16, looser than fp32' if dtype == 'bf16x3' else 'operands rounded to bf16: narrower than the reference\'s fp32'),
    'update_dtype': 'f32 (hand-written convolutions in true float32 on the matrix cores, csrc/train_conv.hip; cross-correlation as bf16x3 split)',
    'update_hand_convs': agent._hand is not None,
    'replay_next_index': 'reference (memory.py:239-242, literal)', 'prefetch': 3,
    'update_early_gradient': agent._early,   # the update's gradient half runs beside the collect step (same results: the minibatch was drawn `prefetch` updates ago)
    'env_groups': groups,   # handles the rank's envs are held as (PipelinedVecStackEnv: a group's forward runs under the other's straggler tail)
    'iterations': iters, 'warmup': warm,
    # placements per second, as leg A counts them (the auto-reset call of every episode is stepped and timed but places
    # nothing); step_calls_per_s counts every vectorised step() call
    'env_steps_per_s': steps_all / dt_max, 'step_calls_per_s': steps_all / dt_max * iters / max(placed_calls, 1),
    'placement_calls': placed_calls, 'iters_per_s': iters / dt_max, 'ms_per_iter': 1e3 * dt_max / iters,
    'rollout_forward_ms': fwd_ms, 'rollout_forward_alone_ms': fwd_alone_ms, 'update_ms': upd_ms, 'allreduce_ms': ar_ms,
    'grad_bucket_bytes': int(agent._flat_grad.numel() * 4), 'update_graphed': agent._train_graph is not None or agent._grad_graph is not None,
    'roofline': {
      'kernel': 'Q-net rollout forward (DeepQSiamFCN, {} samples)'.format(B), 'bound': 'mfma',
      'achieved': flops_fwd / (fwd_ms * 1e-3) / 1e12, 'peak': PEAK_TFLOPS[dtype], 'unit': 'TFLOP/s',
      'frac': flops_fwd / (fwd_ms * 1e-3) / 1e12 / PEAK_TFLOPS[dtype], 'traffic': None, **mfma_counters(dtype),
      'alg_flops_per_launch': flops_fwd, 'avg_launch_ms': fwd_ms,
      # the same forward with nothing else on the device (three evaluations of the batch before the loop)
      'alone': {'avg_launch_ms': fwd_alone_ms, 'achieved': flops_fwd / (fwd_alone_ms * 1e-3) / 1e12,
                'frac': flops_fwd / (fwd_alone_ms * 1e-3) / 1e12 / PEAK_TFLOPS[dtype]},
      'note': 'model-level: 2 x {:.1f} M MAC per sample (SURVEY.md N1) x samples / HIP-event time of the policy evaluations of an '
              'iteration in the timed region (with env_groups > 1 they run beside the other group\'s settle kernel); peak = {}'.format(macs / 1e6, 'dense bf16 MFMA' if dtype == 'bf16' else
                                                   'dense bf16 MFMA / 3 (fp32-class products = three bf16 MFMAs; the fp32 MFMA peak '
                                                   'would be 157.3)'),
    },
  }
  env.close()
  del agent, net, tr
  torch.cuda.empty_cache()
  return out


# ----------------------------------------------------------------------------------------------- worker
def worker(args):
  # stdout carries the ONE JSON line and nothing else: libraries that print banners to stdout (RCCL's version block)
  # are sent to stderr for the whole run, the line is written to the saved descriptor at the end
  sys.stdout.flush()
  json_fd = os.dup(1)
  os.dup2(2, 1)

  def emit(line, provisional=False):
    sys.stdout.flush()
    os.write(json_fd, ((PROVISIONAL if provisional else '') + json.dumps(line) + '\n').encode())

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world != args.gpus:
    raise SystemExit('--gpus {} but WORLD_SIZE={}'.format(args.gpus, world))
  from stackrl_amd.config import SOLVER_PRESETS
  solver_kw = dict(SOLVER_PRESETS[args.solver])
  L = args.rocks or 8

  if args.launch_only:
    import torch
    import torch.distributed as dist
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    dist.barrier()
    dt, placed = aggregate(1.0 + 0.5 * rank, 100 * (rank + 1), world, 'cpu')
    dist.barrier()
    if rank == 0:
      emit({'metric': 'env steps/sec (batched Stack-v0)', 'value': None, 'unit': 'env_steps/s',
            'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup, 'launch_only': True,
            'note': 'rehearsal of spawn + rendezvous + aggregation; no GPU work was done',
            'aggregate_check': {'max_dt': dt, 'sum_placed': placed}})
    dist.destroy_process_group()
    return

  from stackrl_amd import assets
  pool = assets.default_pool()           # synthetic rocks, generator seed 11 (cached)

  cpu = None
  if rank == 0 and args.gpus == 1 and not args.no_cpu:      # the CPU legs belong to the N = 1 run
    from oracle import oracle as _o
    _o.build()
    cpu = cpu_baseline(L, args.seed, solver_kw)     # before any GPU initialisation in this process

  import torch
  import torch.distributed as dist
  dev = 0 if args.same_device else local_rank
  torch.cuda.set_device(dev)
  use_dist = world > 1 or os.environ.get('SRL_BENCH_FORCE_DIST') == '1'   # the latter: rehearse the RCCL calls on one GPU
  if use_dist:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    # the env path has no data-path collective: its barriers and the two scalar reductions of `aggregate` go over gloo
    # (host side), so leg A depends on nothing but the rendezvous; the gradient all-reduce of leg B runs over RCCL in a
    # group of its own (`rccl_group`), created when leg B starts
    dist.init_process_group('gloo', rank=rank, world_size=world)

  def barrier():
    torch.cuda.synchronize()
    if use_dist:
      dist.barrier()
    torch.cuda.synchronize()

  a = env_leg(args, rank, world, pool, barrier, solver_kw)
  dt_max, placed_all = aggregate(a['dt'], a['placed'], world, 'cpu')

  def make_line(dqn, mse):
    B, res, ms, nl, sub, sw = a['B'], a['res'], a['ms'], a['nl'], a['sub'], a['sw']
    traffic, traffic_source = traffic_record(B, a['L'], res, a['alg'] / max(int(nl[1]), 1))
    render_s = float(ms[1]) / 1e3
    cfg = a['config']
    line = {
      'metric': 'env steps/sec (batched Stack-v0)', 'value': placed_all / dt_max, 'unit': 'env_steps/s',
      'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt_max / args.steps,
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',   # leg A computes in float32
      'config': {
        'workload': 'Stack-v0, {} vectorised envs per GPU, {} rocks, random policy on device, physics+render '
                    'kernels only (BASELINE configs[1])'.format(B, a['L']),
        'envs_per_gpu': B, 'episode_length': a['L'], 'heightmap': res, 'sim_time_step': cfg.sim_time_step,
        'solver': args.solver, 'solver_iterations': cfg.solver_iterations, 'warmstart': cfg.warmstart,
        'linear_slop': cfg.linear_slop, 'residual_threshold': cfg.residual_threshold,
        'parallelism': 'env-shard x{}'.format(args.gpus),
        'mesh_pool': '{} synthetic rocks (generator seed 11)'.format(len(pool)),
      },
      'step_calls_per_s': args.steps * B * args.gpus / dt_max,
      # the same loop over 54 calls = six whole episodes (the K-step window above starts mid-episode and weighs the fuller scenes more)
      'value_long': None if a.get('long_value') is None else {'value': a['long_value'], 'steps': a['long_steps'], 'unit': 'env_steps/s'},
      # the same envs as handles that are not kept in lock-step (see env_leg): a capability of the interface, not the contract's figure
      'value_free_running': a.get('free_run'),
      # which build of the env library was timed (stackrl_amd/build.py: 'vectorised+rewritten' = SLP vectoriser + the pass of
      # isa_fix.py over the assembly; 'safe' = the fall-back without the vectoriser) and the hash of the sources it carries
      'env_library': build_info(),
      'roofline': {
        'kernel': 'srl_k_render', 'bound': 'hbm', 'achieved': a['alg'] / render_s / 1e9 if render_s > 0 else None,
        'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': (a['alg'] / render_s / 1e9) / HBM_PEAK_GBS if render_s > 0 else None, 'traffic': traffic,
        'traffic_source': traffic_source,
        'avg_launch_us': 1e3 * float(ms[1]) / max(int(nl[1]), 1), 'launches': int(nl[1]),
        'alg_bytes_per_launch': a['alg'] / max(int(nl[1]), 1),
        # rocks in the scene, averaged over the timed launches (a whole episode cycle: L / 2; the kernel's time grows with it)
        'scene_rocks_mean': a['rocks'] / max(int(nl[1]), 1),
        # the rocks' planes / outlines (the records srl_k_render reads) are made in the TAIL of srl_k_step since round 5: an env's
        # workgroup stages its rocks once its stop criterion has fired (csrc/stage.h) — no kernel between the settle and the
        # render kernel any more (round 4: srl_k_stage, 13.3 us per step, not part of the fraction above); srl_k_stage remains
        # for the explicit-pose hook and is launched on the step path only after a test hook has moved bodies
        'stage': {'where': 'tail of srl_k_step (per env, after its stop criterion)', 'srl_k_stage_launches_in_timed_region': int(nl[2]),
                  'srl_k_stage_avg_launch_us': (1e3 * float(ms[2]) / int(nl[2])) if int(nl[2]) else None},
        'path_note': 'the render path of a step is this one kernel: frac is the path\'s fraction',
      },
      'settle': {
        'kernel': 'srl_k_step', 'avg_launch_ms': float(ms[0]) / max(int(nl[0]), 1), 'launches': int(nl[0]),
        'share_of_wall': float(ms[0]) / 1e3 / a['dt'], 'substeps_mean': float(sub.mean()),
        'substeps_mean_of_per_step_max': float(sub.max(1).mean()), 'substeps_max': int(sub.max()),
        'sweeps_per_substep_mean': float(sw.sum() / max(sub.sum(), 1)),
        'sweeps_mean': float(sw.mean()), 'sweeps_mean_of_per_step_max': float(sw.max(1).mean()),
        'note': 'latency-bound: one launch lasts as long as its slowest env (stop criterion simulator.py:322-335)',
        **settle_counters(B, a['L'], res),
      },
      'cpu_baseline': cpu, 'reward_mse_vs_cpu': mse, 'dqn': dqn,
    }
    if args.config != 1 and dqn:
      # the DQN leg as the headline: the fp32-class rollout (bf16x3 products; the reference computes in fp32) when it was run
      d = dqn.get('bf16x3') or next(iter(dqn.values()))
      line.update(value=d['env_steps_per_s'], steps=d['iterations'], warmup=d['warmup'], ms_per_step=d['ms_per_iter'],
                  dtype=d['rollout_dtype'], env_only={'value': placed_all / dt_max, 'steps': args.steps,
                                                      'warmup': args.warmup, 'ms_per_step': 1e3 * dt_max / args.steps})
      line['config'] = dict(line['config'], workload=d['workload'], note='leg A (config keys above) is reported under env_only')
    return line

  if rank == 0 and world > 1 and os.environ.get('SRL_BENCH_CHILD') == '1' and args.config == 1 and not args.no_dqn:
    # leg A is complete on every rank: hand the headline to the launcher now, so that a rank lost in leg B (whose
    # collectives the others would then wait in) cannot take it down; the final line below replaces it
    emit(make_line({'error': 'a rank ended during leg B (DQN); leg A above is complete'}, None), provisional=True)

  dqn = None
  if not args.no_dqn:
    dqn = {}
    for dtype in (['bf16x3', 'bf16'] if args.rollout == 'both' else ['bf16x3' if args.rollout == 'f32' else args.rollout]):
      # leg B must not take the headline down with it: an exception is recorded, not raised (with the DQN leg as the
      # headline, --config 2|3|4, it is raised)
      try:
        dqn[dtype] = dqn_leg(args, rank, world, pool, barrier, solver_kw, dtype)
      except Exception as e:   # noqa: BLE001
        if args.config != 1 or world > 1:   # with other ranks waiting in this leg's collectives there is no going on
          raise
        import traceback
        traceback.print_exc()
        dqn[dtype] = {'error': '{}: {}'.format(type(e).__name__, e)}
        torch.cuda.synchronize()

  mse = None
  if rank == 0 and args.gpus == 1 and not args.no_cpu:
    mse = reward_mse(args, pool, L, solver_kw)

  if rank == 0:
    emit(make_line(dqn, mse))
  if use_dist:
    dist.barrier()
    dist.destroy_process_group()


def main(argv=None):
  argv = list(sys.argv[1:] if argv is None else argv)
  args = parse_args(argv)
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(launch(args, argv))
  worker(args)


if __name__ == '__main__':
  main()
