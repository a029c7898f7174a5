#!/usr/bin/env python3
"""bench.py — env steps/sec of the batched Stack-v0 hot path (BASELINE.json metric).

A "step" is one vectorised `env.step` over the per-GPU batch (settle kernel + render kernel).
N = 1 workload = BASELINE.json configs[1]: Stack-v0, 1024 vectorised envs, 8 rocks, random policy on
device, physics + render kernels only.  N > 1: every rank owns its own 1024 envs (weak scaling; envs are
independent, utils.py:424-448, so there is no data-path collective).

`value` = placements performed by all ranks / max-over-ranks wall time of the K timed steps (auto-reset
calls, env.py:235-236, are stepped and timed but are not placements).  Inputs are resident in HBM.

Extra objects on the JSON line: `roofline` (render kernel K2, HBM-bound: algorithmic bytes / HIP-event
time), `settle` (K1 launch statistics), `cpu_baseline` (the CPU oracle timed on this
host's cores on a bounded sample of the same workload), `reward_mse_vs_cpu`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)


def alg_bytes_per_env(res, r, nb):
  """H f32 + obs u8 (H,W,2) + O f32 + obs u8 (h,w,1) + 1,404 B per placed rock."""
  return 6 * res * res + 5 * r * r + 1404 * nb


def _cpu_worker(args):
  n, L, seed, offset, episodes = args
  from oracle.oracle import OracleEnv
  from stackrl_amd import assets
  from stackrl_amd.config import StackConfig
  pool = assets.default_pool()
  env = OracleEnv(StackConfig(n_envs=n, episode_length=L, env_index_offset=offset), pool, seed=seed)
  env.reset()
  t0 = time.perf_counter()
  for _ in range(episodes * L):
    env.step(env.sample())
  return n * L * episodes, time.perf_counter() - t0


def cpu_baseline(L, seed, budget_envs=64, episodes=3):
  """The oracle on this host's cores: `cores` processes x `budget_envs` envs x `episodes` episodes."""
  import multiprocessing as mp
  cores = min(os.cpu_count() or 1, 16)
  # one-thread figure first (same sample size per worker)
  n1, t1 = _cpu_worker((budget_envs, L, seed, 0, episodes))
  ctx = mp.get_context('fork')    # forked before this process touches the GPU
  t0 = time.perf_counter()
  with ctx.Pool(cores) as pool:
    res = pool.map(_cpu_worker, [(budget_envs, L, seed, i * budget_envs, episodes) for i in range(cores)])
  wall = time.perf_counter() - t0
  placed = sum(r[0] for r in res)
  busy = max(r[1] for r in res)
  return {
    'value': placed / busy, 'unit': 'env_steps/s', 'cores': cores, 'kind': 'port',
    'sample': '{} procs x {} envs x {} episode(s) of {} placements (oracle/srl_oracle.c, same pool and RNG keys); '
              'wall incl. process start {:.1f}s'.format(cores, budget_envs, episodes, L, wall),
    'single_thread_value': n1 / t1,
  }


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


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=54)
  ap.add_argument('--warmup', type=int, default=9)
  ap.add_argument('--envs', type=int, default=1024, help='envs per GPU')
  ap.add_argument('--rocks', type=int, default=8, help='episode_length')
  ap.add_argument('--seed', type=int, default=11)
  ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
  args = ap.parse_args()

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit('--gpus {} needs torch.distributed.run with --nproc-per-node {}'.format(args.gpus, args.gpus))
  B, L = args.envs, args.rocks

  from stackrl_amd import assets
  pool = assets.default_pool()           # synthetic rocks, generator seed 11 (cached)

  cpu = None
  if rank == 0 and args.gpus == 1 and not args.no_cpu:
    from oracle import oracle as _o
    _o.build()
    cpu = cpu_baseline(L, args.seed)     # before any GPU initialisation in this process

  import torch
  import torch.distributed as dist
  torch.cuda.set_device(local_rank)
  use_dist = world > 1 or os.environ.get('SRL_BENCH_FORCE_DIST') == '1'   # the latter: rehearse the RCCL calls on one GPU
  if use_dist:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

  from stackrl_amd import env as envs
  env = envs.VecStackEnv(n_parallel=B, seed=args.seed, pool=pool, block=False, episode_length=L,
                         env_index_offset=rank * B)

  def barrier():
    torch.cuda.synchronize()
    if use_dist:
      dist.barrier()
    torch.cuda.synchronize()

  # lock-step bookkeeping: which calls are placements and how many rocks each render launch sees
  phase = {'k': 0}   # calls since reset(): 1..L placements, L+1 auto-reset

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
  placed, alg = 0, 0
  res, r = env.config.overhead_res, env.config.object_res
  last = None
  for _ in range(args.steps):
    last, p, nb = do_step()
    placed += p
    alg += B * alg_bytes_per_env(res, r, nb)
  barrier()
  dt = time.perf_counter() - t0
  last()     # raises if any env diverged / action invalid
  ms, nl = env.kernel_times()
  env.set_profiling(False)
  # sub-step statistics of one further (untimed) episode: what the stop criterion asked of the settle kernel
  subs = []
  while phase['k'] != 0:
    do_step()[0]()
  for _ in range(L):
    do_step()[0]()
    subs.append(env.state()[2].sum(1))
  do_step()[0]()
  sub = np.stack(subs)

  dt_max, placed_all = aggregate(dt, placed, world, 'cuda')

  mse = None
  if rank == 0 and args.gpus == 1 and not args.no_cpu:
    # reward parity on identical seeds: 64 envs, one episode
    from oracle.oracle import OracleEnv
    from stackrl_amd.config import StackConfig
    g = envs.VecStackEnv(n_parallel=64, seed=args.seed + 1, pool=pool, block=True, episode_length=L)
    o = OracleEnv(StackConfig(n_envs=64, episode_length=L), pool, seed=args.seed + 1)
    g.reset(); o.reset()
    se, cnt, idx_ok = 0.0, 0, True
    for _ in range(L):
      a = g.sample()
      (gm, go), gr, gd = g.step(a)
      (om, oo), orr, od = o.step(a.cpu().numpy())
      se += float(((gr.cpu().numpy().astype(np.float64) - orr) ** 2).sum()); cnt += 64
      idx_ok &= bool(np.array_equal(gm.cpu().numpy(), om) and np.array_equal(gd.cpu().numpy(), od))
    mse = {'value': se / cnt, 'envs': 64, 'steps': L, 'obs_and_done_bit_exact': idx_ok}
    g.close()

  traffic = None
  # HBM bytes per launch from the PMC counters: separate rocprofv3 --pmc passes of this command (they cannot run inside
  # this process), summarised by tools/pmc_summary.py into profiles/rNN_render_pmc.json; the latest round's file is used
  import glob
  pmcs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_render_pmc.json')))
  if pmcs and B == 1024 and L == 8:
    with open(pmcs[-1]) as f:
      traffic = json.load(f).get('traffic_bytes_per_launch')
  if rank == 0:
    render_s = float(ms[1]) / 1e3
    line = {
      'metric': 'env steps/sec (batched Stack-v0)', 'value': placed_all / dt_max, 'unit': 'env_steps/s',
      'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt_max / args.steps,
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {
        'workload': 'Stack-v0, {} vectorised envs per GPU, {} rocks, random policy on device, physics+render '
                    'kernels only (BASELINE configs[1])'.format(B, L),
        'envs_per_gpu': B, 'episode_length': L, 'heightmap': res, 'sim_time_step': env.config.sim_time_step,
        'solver_iterations': env.config.solver_iterations, 'parallelism': 'env-shard x{}'.format(args.gpus),
        'mesh_pool': '{} synthetic rocks (generator seed 11)'.format(len(pool)),
      },
      'step_calls_per_s': args.steps * B * args.gpus / dt_max,
      'roofline': {
        'kernel': 'srl_k_render', 'bound': 'hbm', 'achieved': alg / render_s / 1e9 if render_s > 0 else None,
        'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': (alg / render_s / 1e9) / HBM_PEAK_GBS if render_s > 0 else None, 'traffic': traffic,
        'avg_launch_us': 1e3 * float(ms[1]) / max(int(nl[1]), 1), 'launches': int(nl[1]),
        'alg_bytes_per_launch': alg / max(int(nl[1]), 1),
      },
      'settle': {
        'kernel': 'srl_k_step', 'avg_launch_ms': float(ms[0]) / max(int(nl[0]), 1), 'launches': int(nl[0]),
        'share_of_wall': float(ms[0]) / 1e3 / dt, 'substeps_mean': float(sub.mean()),
        'substeps_mean_of_per_step_max': float(sub.max(1).mean()), 'substeps_max': int(sub.max()),
        'note': 'latency-bound: one launch lasts as long as its slowest env (stop criterion simulator.py:322-335)',
      },
      'cpu_baseline': cpu, 'reward_mse_vs_cpu': mse,
    }
    print(json.dumps(line))
  env.close()
  if use_dist:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
