"""Two-rank rehearsal of the multi-GPU update path on ONE GPU (runs first, by file name: the ranks are forked from a
pytest process that has not touched the GPU yet — nothing is exec'd, and a process that has initialised HIP is never
forked)."""
import math

import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu


def _two_rank_update(rank, world, port, out):
  """One rank of the two-rank rehearsal on ONE GPU (both ranks on device 0, gloo): different net seeds and different
  replay contents per rank; the update runs eagerly first, then from the two hipGraphs around the all-reduce."""
  import os
  import torch.distributed as dist
  from stackrl_amd import nets
  from stackrl_amd.dqn import DQN
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.cuda.set_device(0)
  net = nets.DeepQSiamFCN(seed=10 + rank).cuda()               # DQN broadcasts rank 0's weights
  B = 4
  agent = DQN(net, learning_rate=1e-3, adam_betas=(0.95, 0.95), minibatch_size=4, replay_memory_size=B * 8,
              discount_factor=0.9, collect_batch_size=B, exploration=0.5, prioritization=0.6,
              priority_bias_compensation=0.5, double=True, seed=100 + rank, xcorr='bf16x3', graphs=True)
  g = torch.Generator(device='cuda').manual_seed(50 + rank)
  for t in range(7):
    st = (torch.randint(0, 256, (B, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8),
          torch.randint(0, 256, (B, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8))
    agent.observe(st, torch.randn(B, generator=g, device='cuda'), torch.zeros(B, dtype=torch.bool, device='cuda'),
                  torch.randint(0, net.n_actions, (B,), generator=g, device='cuda'))
  w0 = torch.cat([p.detach().flatten() for p in agent._params]).clone()
  losses = [float(agent.train()[0]) for _ in range(DQN._GRAPH_WARMUP + 3)]
  flat = torch.cat([p.detach().flatten() for p in agent._params]).cpu()
  gathered = [torch.zeros_like(flat) for _ in range(world)]
  dist.all_gather(gathered, flat)
  lo = torch.tensor(losses); lall = [torch.zeros_like(lo) for _ in range(world)]
  dist.all_gather(lall, lo)
  if rank == 0:
    torch.save({'replicas_equal': bool(torch.equal(gathered[0], gathered[1])), 'graphed': agent._train_graph is not None
                and getattr(agent, '_apply_graph', None) is not None, 'moved': bool(not torch.equal(flat, w0.cpu())),
                'losses_differ': bool(not torch.equal(lall[0], lall[1])), 'finite': bool(torch.isfinite(lo).all())}, out)
  dist.destroy_process_group()


def test_two_rank_graphed_update_keeps_replicas_equal(tmp_path):
  """Multi-GPU update path on one GPU: two processes (gloo, both on device 0), nets built from different seeds, different
  replay shards; after eager and hipGraph-replayed updates (two graphs around the all-reduce of the gradient bucket) the
  replicas hold bit-identical weights while their local losses differ."""
  import os
  import torch.multiprocessing as mp
  if torch.cuda.is_initialized():
    pytest.skip('this process has already initialised the GPU: its children must not be forked from it')
  out = str(tmp_path / 'res.pt')
  mp.start_processes(_two_rank_update, args=(2, 29000 + os.getpid() % 2000, out), nprocs=2, join=True, start_method='fork')
  res = torch.load(out)
  assert res == {'replicas_equal': True, 'graphed': True, 'moved': True, 'losses_differ': True, 'finite': True}


def _run_bench(extra_env, *flags):
  import json, os, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, **extra_env)
  env.pop('RANK', None); env.pop('WORLD_SIZE', None)
  p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--same-device', '--backend', 'gloo',
                      '--no-cpu', '--envs', '64', '--steps', '3', '--warmup', '1', '--dqn-envs', '32', '--dqn-rocks', '4',
                      '--dqn-iters', '3', '--dqn-warmup', '1', '--dqn-slots', '8'] + list(flags),
                     env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
  lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
  return p.returncode, lines, p.stderr.decode()


def test_bench_two_ranks_one_json_line_and_survives_a_lost_rank():
  """`python bench.py --gpus 2` as the driver runs it (here: both ranks on device 0, gloo): the launcher starts the
  ranks, exactly one JSON line comes out, leg B carries the all-reduce time; and when a rank is lost in leg B the
  launcher still prints leg A's aggregated headline, marked as such, ends the other rank and returns non-zero."""
  import json
  if torch.cuda.is_initialized():
    pytest.skip('this process has already initialised the GPU: it must not start (fork + exec) other programs')
  rc, lines, err = _run_bench({})
  assert rc == 0 and len(lines) == 1, err[-2000:]
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['value'] > 0 and d['config']['parallelism'] == 'env-shard x2'
  assert set(d['dqn']) == {'bf16x3', 'bf16'} and all(v['allreduce_ms'] is not None and v['update_graphed'] for v in d['dqn'].values())
  rc, lines, err = _run_bench({'SRL_BENCH_FAIL_LEG_B': '1'})
  assert rc != 0 and len(lines) == 1, err[-2000:]      # the headline is printed, but a lost rank is not a success
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['value'] > 0 and 'error' in d['dqn']
