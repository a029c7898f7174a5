"""Host-side mirror of the reference's batched env interface over the HIP C-ABI.

`VecStackEnv` is what `stackrl.envs.make('Stack-v0', n_parallel=B, seed=s, **kwargs)` returns in
the reference (`stackrl/envs/utils.py:44-141` -> `ParallelEnv`, `:302-576`): same method and property
names (`step/reset/sample/seed/close/__call__`, `batch_size`, `observation_spec`, `action_spec`,
`multiprocessing`), same argument meaning, same tuple layout
`((u8[B,H,W,2], u8[B,h,w,1]), f32[B], bool[B])`, same non-blocking default (a callable that yields the
time step, `utils.py:468-486`), same exception types.  Tensors are torch (ROCm) instead of tf.

All arithmetic happens in libstackrl_hip.so; torch only owns device memory and streams.
"""
import collections
import ctypes
import os

import numpy as np
import torch

from stackrl_amd import assets as _assets
from stackrl_amd import lib as _lib
from stackrl_amd import config as _config
from stackrl_amd.config import StackConfig

TensorSpec = collections.namedtuple('TensorSpec', ['shape', 'dtype'])


def _check(rc):
  if rc == _config.OK:
    return
  msg = _lib.last_error()
  if rc == _config.EINVAL_ACTION:
    raise AssertionError(msg)             # env.py:238
  if rc == _config.ESIM_DIVERGED:
    raise RuntimeError(msg)               # simulator.py:221-224
  if rc in (_config.EINVAL, _config.ENOMESH):
    raise ValueError(msg)
  raise RuntimeError('HIP error: ' + msg)


def _np_ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class VecStackEnv(object):
  """B independent Stack-v0 envs on one GPU (drop-in for `ParallelEnv`, utils.py:302)."""

  def __init__(self, n_parallel=None, block=None, seed=None, pool=None, device=None,
               env_index_offset=0, side_stream=False, concurrent_envs=None, stream_priority=None, launch_order=None,
               **kwargs):
    """
    Args:
      n_parallel: number of environments B (utils.py:324).
      block: whether step/reset block by default; None -> False (utils.py:326-327).
      seed: env i uses (seed + env_index_offset + i) % 2**32 (utils.py:433).
      pool: `assets.MeshPool` (the urdf list of env.py:92-103); None -> synthetic default pool.
      device: torch device (defaults to the current cuda device).
      env_index_offset: global index of env 0 when the batch is sharded over ranks.
      side_stream: run the env kernels on their own HIP stream so that a non-blocking `step` overlaps the
        caller's work on the current stream (the reference overlaps env processes with `agent.train()`,
        training.py:359-368); the returned callable joins the streams.
      stream_priority: priority of the side stream (-1 = high: the dispatcher serves the env's long-running workgroups
        before the kernels of the current stream; None / 0 = default).
      concurrent_envs: envs that step on the device at the same time over all handles of the caller (a tuning hint for
        the settle kernel's build, `srl_set_concurrent_envs`; results do not depend on it).  None -> this handle alone.
      launch_order: order in which the settle kernel's workgroups take the envs (`srl_set_launch_order`): None -> by batch
        size, False -> index order, True -> the envs with the highest release first.  Results do not depend on it.
      kwargs: StackEnv arguments (env.py:28-51), e.g. episode_length, sim_time_step, rewarder.
    """
    if not torch.cuda.is_available():
      raise RuntimeError('VecStackEnv needs a HIP device: there is no CPU fallback in the product path.')
    self._lib = _lib.load()
    self._device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    self.config = StackConfig(n_envs=int(n_parallel or 1), env_index_offset=int(env_index_offset), **kwargs)
    self._block = block
    self._c = self.config.to_c()
    self._h = ctypes.c_void_p()
    with torch.cuda.device(self._device):
      _check(self._lib.srl_create(ctypes.byref(self._c), ctypes.byref(self._h)))
      if concurrent_envs:
        _check(self._lib.srl_set_concurrent_envs(self._h, int(concurrent_envs)))
      if launch_order is not None:
        _check(self._lib.srl_set_launch_order(self._h, int(bool(launch_order))))
      self.pool = pool if pool is not None else _assets.default_pool()
      p = self.pool
      _check(self._lib.srl_load_meshes(self._h, _np_ptr(p.verts), _np_ptr(p.vert_off), _np_ptr(p.tris),
                                       _np_ptr(p.tri_off), _np_ptr(p.mass_com), len(p)))
    B, H, h = self.config.n_envs, self.config.overhead_res, self.config.object_res
    # TestStackEnv (env.py:443-470): one object map per observable orientation, action = orientation * A + pixel
    self._no = self.config.n_object_maps   # with ordering freedom: the maps of every rock of the episode (empty once placed)
    self._observation_spec = (TensorSpec((H, H, 2), torch.uint8),
                              TensorSpec((h, h, 1) if self._no == 1 else (self._no, h, h, 1), torch.uint8))
    self._action_spec = TensorSpec((), torch.int64)
    self._B, self._H, self._hh = B, H, h
    self._closed = False
    self._left = 0          # rocks of the running episode not yet placed (host mirror of the lock-step episode machine)
    self._side = torch.cuda.Stream(device=self._device, priority=int(stream_priority or 0)) if side_stream else None
    self.seed(seed if seed is not None else 0)

  # ---- properties (utils.py:270-281, :417-422)
  @property
  def multiprocessing(self):
    return False

  @property
  def batch_size(self):
    return self._B

  @property
  def observation_spec(self):
    return self._observation_spec

  @property
  def action_spec(self):
    return self._action_spec

  @property
  def n_actions(self):
    return self.config.n_actions * self._no

  @property
  def num_maps_on_show(self):
    """Object maps that hold a rock in the latest observation (`Observer.num_objects`, observer.py:370-376, as
    env.py:596-608 uses it to size the spaces): an action's index must be below this.  Every env of the batch is at
    the same point of its episode, so this is one number (valid once the latest step has been waited for)."""
    if self.config.ordering_freedom:
      return self._left * self.config.n_orientations
    return self.config.n_orientations if self._left > 0 else 0

  def __call__(self, *args, **kwargs):
    return self.step(*args, **kwargs)

  def __del__(self):
    try:
      self.terminate()
    except Exception:
      pass

  # ---- lifecycle
  def terminate(self):
    if getattr(self, '_h', None) and not self._closed:
      self._closed = True
      self._lib.srl_destroy(self._h)
      self._h = None

  def close(self):
    self.terminate()

  def seed(self, seed):
    """utils.py:522-532: returns the list of per-env seeds."""
    seed = int(seed) % 2**32
    _check(self._lib.srl_seed(self._h, seed))
    off = self.config.env_index_offset
    return [[(seed + off + i) % 2**32] for i in range(self._B)]

  def set_script(self, mesh_ids, goal_rect):
    """Explicit episode script for the next reset (parity harness)."""
    mesh_ids = np.ascontiguousarray(mesh_ids, np.int32)
    goal_rect = np.ascontiguousarray(goal_rect, np.int32)
    if mesh_ids.shape != (self._B, self.config.episode_length) or goal_rect.shape != (self._B, 4):
      raise ValueError('script shapes must be [B, L] and [B, 4]')
    _check(self._lib.srl_set_script(self._h, _np_ptr(mesh_ids), _np_ptr(goal_rect)))

  # ---- stepping
  def _stream(self):
    st = self._side if self._side is not None else torch.cuda.current_stream(self._device)
    return ctypes.c_void_p(st.cuda_stream)

  def _fork(self):
    if self._side is not None:      # env kernels start after everything already queued by the caller (the action)
      self._side.wait_stream(torch.cuda.current_stream(self._device))

  def _join(self):
    if self._side is not None:
      torch.cuda.current_stream(self._device).wait_stream(self._side)

  def _new_obs(self):
    return (torch.empty((self._B, self._H, self._H, 2), dtype=torch.uint8, device=self._device),
            torch.empty((self._B,) + tuple(self._observation_spec[1].shape), dtype=torch.uint8, device=self._device))

  def _finish(self, out):
    def wait():
      _check(self._lib.srl_sync_status(self._h, self._stream()))
      self._join()
      return out
    return wait

  def reset(self, block=None, out=None):
    """out: optional (obs_map, obs_obj) tensors to write into (views of a larger batch, `PipelinedVecStackEnv`)."""
    om, oo = self._new_obs() if out is None else out
    self._fork()
    with torch.cuda.device(self._device):
      _check(self._lib.srl_reset(self._h, om.data_ptr(), oo.data_ptr(), self._stream()))
    self._left = self.config.episode_length
    if self._side is not None:      # the caching allocator must not hand these blocks out again while the side stream writes them
      for t in (om, oo):
        t.record_stream(self._side)
    keys = self.config.reward_keys
    out = ((om, oo), torch.zeros(self._B if keys is None else (self._B, len(keys)), dtype=torch.float32, device=self._device),
           torch.zeros(self._B, dtype=torch.bool, device=self._device))     # utils.py:545-552
    wait = self._finish(out)
    block = self._block if block is None else block
    return wait() if block else wait

  def step(self, action, block=None, out=None):
    """out: optional (obs_map, obs_obj, reward, done uint8) tensors to write into (contiguous views of a larger batch)."""
    if not torch.is_tensor(action):
      action = torch.as_tensor(action)
    action = action.to(device=self._device, dtype=torch.int64).contiguous()
    if action.shape != (self._B,):
      raise ValueError('action must have shape [{}]'.format(self._B))
    keys = self.config.reward_keys       # 'all' / 'eval': one column per key of the reference's dict (rewarder.py:147-158)
    if out is None:
      om, oo = self._new_obs()
      reward = torch.empty(self._B if keys is None else (self._B, len(keys)), dtype=torch.float32, device=self._device)
      done = torch.empty(self._B, dtype=torch.uint8, device=self._device)
    else:
      om, oo, reward, done = out
    self._fork()
    with torch.cuda.device(self._device):
      _check(self._lib.srl_step(self._h, action.data_ptr(), om.data_ptr(), oo.data_ptr(), reward.data_ptr(),
                                done.data_ptr(), self._stream()))
    self._left = self.config.episode_length if self._left == 0 else self._left - 1   # env.py:235-236: auto-reset
    if self._side is not None:
      for t in (om, oo, reward, done, action):
        t.record_stream(self._side)
    out = ((om, oo), reward, done.view(torch.bool))
    self._keep = action   # keep the action tensor alive until the kernels consumed it
    wait = self._finish(out)
    block = self._block if block is None else block
    return wait() if block else wait

  def sample(self):
    """utils.py:534-538: a batch of uniform random actions."""
    a = torch.empty(self._B, dtype=torch.int64, device=self._device)
    self._fork()
    with torch.cuda.device(self._device):
      _check(self._lib.srl_sample(self._h, a.data_ptr(), self._stream()))
    if self._side is not None:
      a.record_stream(self._side)
    self._join()
    return a

  # ---- telemetry (Simulator.poses / n_steps, Observer.state, Rewarder.goal)
  def state(self):
    B = self._B
    poses = np.zeros((B, _config.MAX_BODIES, 8), np.float32)
    nb = np.zeros(B, np.int32)
    sub = np.zeros((B, 2), np.int32)
    st = np.zeros(B, np.int32)
    _check(self._lib.srl_get_state(self._h, _np_ptr(poses), _np_ptr(nb), _np_ptr(sub), _np_ptr(st)))
    return poses, nb, sub, st

  def velocities(self):
    v = np.zeros((self._B, _config.MAX_BODIES, 8), np.float32)
    _check(self._lib.srl_get_velocities(self._h, _np_ptr(v)))
    return v

  def set_body_state(self, poses=None, velocities=None):
    """`resetBasePositionAndOrientation` / `resetBaseVelocity` (simulator.py:313, :214) for every placed body, in the
    layouts of `state()[0]` / `velocities()`.  Test hook of the closed-form physics cases."""
    p = None if poses is None else np.ascontiguousarray(poses, np.float32)
    v = None if velocities is None else np.ascontiguousarray(velocities, np.float32)
    for a in (p, v):
      if a is not None and a.shape != (self._B, _config.MAX_BODIES, 8):
        raise ValueError('expected an array of shape ({}, {}, 8)'.format(self._B, _config.MAX_BODIES))
    _check(self._lib.srl_set_body_state(self._h, None if p is None else _np_ptr(p), None if v is None else _np_ptr(v)))

  def step_simulation(self, n=1):
    """`pb.stepSimulation` x n on every env (no placement, no stop criterion, no render)."""
    _check(self._lib.srl_step_simulation(self._h, int(n), self._stream()))
    _check(self._lib.srl_sync_status(self._h, self._stream()))

  def sweeps(self):
    """Solver sweeps run by the last step of each env (telemetry)."""
    sw = np.zeros(self._B, np.int32)
    _check(self._lib.srl_get_sweeps(self._h, _np_ptr(sw)))
    return sw

  def contacts(self):
    mp = np.zeros(self._B, np.float32)
    npts = np.zeros(self._B, np.int32)
    _check(self._lib.srl_get_contacts(self._h, _np_ptr(mp), _np_ptr(npts)))
    return mp, npts

  def maps(self):
    Hm = np.zeros((self._B, self._H, self._H), np.float32)
    Om = np.zeros((self._B, self._hh, self._hh) if self._no == 1 else (self._B, self._no, self._hh, self._hh), np.float32)
    g = np.zeros((self._B, 4), np.int32)
    _check(self._lib.srl_get_maps(self._h, _np_ptr(Hm), _np_ptr(Om), _np_ptr(g)))
    return Hm, Om, g

  def object_map(self, mesh_id):
    k = self.config.n_orientations
    o = np.zeros((self._hh, self._hh) if k == 1 else (k, self._hh, self._hh), np.float32)
    _check(self._lib.srl_get_object_map(self._h, int(mesh_id), _np_ptr(o)))
    return o

  def render_heightmap(self, poses, mesh_ids, n_bodies, out=None):
    """O1 on explicit poses: poses f32[B,32,7], mesh_ids i32[B,32], n_bodies i32[B] (device tensors)."""
    if out is None:
      out = torch.empty((self._B, self._H, self._H), dtype=torch.float32, device=self._device)
    with torch.cuda.device(self._device):
      _check(self._lib.srl_render_heightmap(self._h, poses.data_ptr(), mesh_ids.data_ptr(), n_bodies.data_ptr(),
                                            out.data_ptr(), self._stream()))
    return out

  def set_profiling(self, enable=True):
    _check(self._lib.srl_set_profiling(self._h, int(bool(enable))))

  def stage_records(self):
    """Test hook (`srl_get_stage_records`): int32 view [B, L, stride, 4] of the rocks' staged render records."""
    stride = int(self._lib.srl_stage_record_stride())
    out = np.zeros((self._B, self.config.episode_length, stride, 4), np.float32)
    _check(self._lib.srl_get_stage_records(self._h, _np_ptr(out), out.size))
    return out

  def kernel_times(self):
    ms = np.zeros(3, np.float32)
    n = np.zeros(3, np.int32)
    _check(self._lib.srl_get_kernel_times(self._h, _np_ptr(ms), _np_ptr(n)))
    return ms, n

  def order_kernel_times(self):
    """(milliseconds, launches) of the ordered launch's two kernels since the last call; call `kernel_times` first."""
    ms = np.zeros(1, np.float32)
    n = np.zeros(1, np.int32)
    _check(self._lib.srl_get_order_kernel_times(self._h, _np_ptr(ms), _np_ptr(n)))
    return float(ms[0]), int(n[0])

  def launch_order(self):
    """Test hook (`srl_get_launch_order`): (keys uint64 [B], order int32 [B]) of the latest ordered launch."""
    keys = np.zeros(self._B, np.uint64)
    order = np.zeros(self._B, np.int32)
    _check(self._lib.srl_get_launch_order(self._h, _np_ptr(keys), _np_ptr(order)))
    return keys, order


class PipelinedVecStackEnv(object):
  """The same B envs as `VecStackEnv(n_parallel=B)` — same seeds, same trajectories, bit for bit — held as `groups`
  handles of B / groups envs, each on its own HIP stream, so that the training loop can treat the groups the way the
  reference's `ParallelEnv` treats its worker processes (utils.py:468-486: every worker is sent its action and answers
  when it is done): `collect_step` evaluates the policy on group k as soon as group k's step has finished and starts
  group k's next step as soon as its actions exist, while the other groups are still settling.

  Why it pays on MI355X: a settle launch lasts as long as its slowest env and for most of
  that time most of its workgroups have finished — at 4,096 envs x 16 rocks about half of the launch's slot-time is
  idle — but the Q-net forward of the NEXT step cannot use those CUs because it needs the observations of every env.
  With two groups the forward of one group runs under the straggler tail of the other.  (More than two groups lose
  again: concurrent settle launches share the CUs round robin, so they all finish late, and HIP multiplexes streams
  over four hardware queues.)

  The plain `ParallelEnv` surface (`reset`, `step`, `sample`, `seed`, ...) is here too; outputs are full-batch tensors
  the groups write their slices of (no copies)."""

  def __init__(self, n_parallel=None, groups=2, block=None, seed=None, pool=None, device=None, env_index_offset=0,
               **kwargs):
    # groups: a count (equal groups; two is the measured optimum, unequal splits of 4,096 envs lose 3 - 6 %) or the sizes
    B = int(n_parallel or 1)
    if isinstance(groups, (tuple, list)):          # explicit group sizes
      sizes = [int(g) for g in groups]
      if any(g < 1 for g in sizes) or sum(sizes) != B:
        raise ValueError('group sizes must be positive and add up to n_parallel')
    else:
      K = int(groups)
      if K < 1 or B % K:
        raise ValueError('n_parallel must be a multiple of groups')
      sizes = [B // K] * K
    K = len(sizes)
    kwargs.pop('side_stream', None)
    kwargs.setdefault('concurrent_envs', B)
    self._block = block
    self._B, self._K = B, K
    self._start = [sum(sizes[:k]) for k in range(K + 1)]
    self.pool = pool if pool is not None else _assets.default_pool()
    self._envs = [VecStackEnv(n_parallel=sizes[k], block=False, seed=seed, pool=self.pool, device=device,
                              env_index_offset=int(env_index_offset) + self._start[k], side_stream=True, **kwargs)
                  for k in range(K)]
    e0 = self._envs[0]
    self.config = e0.config
    self._device = e0._device
    self._cur = None        # the latest step: full-batch tensors + the per-group waits not yet taken

  # ---- the ParallelEnv surface (utils.py:270-281, :417-422)
  multiprocessing = False

  @property
  def batch_size(self):
    return self._B

  @property
  def groups(self):
    return self._K

  @property
  def observation_spec(self):
    return self._envs[0].observation_spec

  @property
  def action_spec(self):
    return self._envs[0].action_spec

  @property
  def n_actions(self):
    return self._envs[0].n_actions

  @property
  def num_maps_on_show(self):
    return self._envs[0].num_maps_on_show

  def __call__(self, *args, **kwargs):
    return self.step(*args, **kwargs)

  def seed(self, seed):
    out = []
    for e in self._envs:
      out += e.seed(seed)
    return out

  def set_script(self, mesh_ids, goal_rect):
    for k, e in enumerate(self._envs):
      e.set_script(mesh_ids[self._slice(k)], goal_rect[self._slice(k)])

  def close(self):
    for e in self._envs:
      e.close()

  terminate = close

  def sample(self):
    return torch.cat([e.sample() for e in self._envs])

  def sweeps(self):
    return np.concatenate([e.sweeps() for e in self._envs])

  def state(self):
    return tuple(np.concatenate(x) for x in zip(*[e.state() for e in self._envs]))

  def maps(self):
    return tuple(np.concatenate(x) for x in zip(*[e.maps() for e in self._envs]))

  def set_profiling(self, enable=True):
    for e in self._envs:
      e.set_profiling(enable)

  def kernel_times(self):
    ms, n = zip(*[e.kernel_times() for e in self._envs])
    return np.sum(ms, 0), np.sum(n, 0)

  def _outputs(self):
    e0 = self._envs[0]
    keys = self.config.reward_keys
    return dict(
      om=torch.empty((self._B,) + tuple(e0._observation_spec[0].shape), dtype=torch.uint8, device=self._device),
      oo=torch.empty((self._B,) + tuple(e0._observation_spec[1].shape), dtype=torch.uint8, device=self._device),
      reward=torch.empty(self._B if keys is None else (self._B, len(keys)), dtype=torch.float32, device=self._device),
      done=torch.empty(self._B, dtype=torch.uint8, device=self._device), waits=[None] * self._K)

  def _slice(self, k):
    return slice(self._start[k], self._start[k + 1])

  def _step_tuple(self, o, s=slice(None)):
    return (o['om'][s], o['oo'][s]), o['reward'][s], o['done'][s].view(torch.bool)

  def _wait_group(self, o, k):
    w, o['waits'][k] = o['waits'][k], None
    if w is not None:
      w()                     # srl_sync_status of the group's handle + the current stream waits for the group's stream

  def drain(self):
    """Wait for every step still in flight (before a reset, a checkpoint, the end of a run)."""
    if self._cur is not None:
      for k in range(self._K):
        self._wait_group(self._cur, k)

  def _waiter(self, o):
    def wait():
      for k in range(self._K):
        self._wait_group(o, k)
      return self._step_tuple(o)
    return wait

  def reset(self, block=None):
    self.drain()
    o = self._outputs()
    o['reward'].zero_(); o['done'].zero_()                                # utils.py:545-552
    for k, e in enumerate(self._envs):
      s = self._slice(k)
      o['waits'][k] = e.reset(block=False, out=(o['om'][s], o['oo'][s]))
    self._cur = o
    wait = self._waiter(o)
    block = self._block if block is None else block
    return wait() if block else wait

  def step(self, action, block=None):
    if not torch.is_tensor(action):
      action = torch.as_tensor(action)
    action = action.to(device=self._device, dtype=torch.int64).contiguous()
    if action.shape != (self._B,):
      raise ValueError('action must have shape [{}]'.format(self._B))
    self.drain()
    o = self._outputs()
    for k, e in enumerate(self._envs):
      s = self._slice(k)
      o['waits'][k] = e.step(action[s], block=False, out=(o['om'][s], o['oo'][s], o['reward'][s], o['done'][s]))
    self._cur = o
    wait = self._waiter(o)
    block = self._block if block is None else block
    return wait() if block else wait

  def collect_step(self, policy):
    """`action = policy(step); env.step(action)` of the training loop (training.py:352-357), group by group: the policy
    call on group k waits for group k's latest step only, and group k's next step starts as soon as its actions exist.
      policy(k, index_slice, (state_k, reward_k, terminal_k)) -> int64 actions [B / groups] of group k
    Returns `(step, action)` of the whole batch: the step the policy has just acted on (complete: every group has been
    waited for) and the actions now being carried out.  Nothing is returned to wait on — the next `collect_step`,
    `step`, `reset` or `drain` takes the waits."""
    if self._cur is None:
      raise RuntimeError('collect_step needs a reset first')
    cur, nxt = self._cur, self._outputs()
    action = torch.empty(self._B, dtype=torch.int64, device=self._device)
    for k, e in enumerate(self._envs):
      s = self._slice(k)
      self._wait_group(cur, k)
      action[s] = policy(k, s, self._step_tuple(cur, s))
      nxt['waits'][k] = e.step(action[s], block=False, out=(nxt['om'][s], nxt['oo'][s], nxt['reward'][s], nxt['done'][s]))
    self._cur = nxt
    return self._step_tuple(cur), action


class StartedVecStackEnv(VecStackEnv):
  """`StartedStackEnv` (Stack-v1, env.py:348-441): an episode uses `n_objects` rocks, the first `n_objects -
  episode_length` of which are placed by `start_policy` inside `reset`, so the agent sees only the last
  `episode_length` placements.  The default start policy is the reference's: the lowest placement whose footprint
  lies fully inside the goal (env.py:391-417), evaluated by the heuristic kernels (csrc/heuristics.hip).
  The batch runs in lock step (same episode length everywhere), so the auto-reset call of `step` (env.py:235-236)
  performs the start placements for all envs at once.  With `min_episode_length` every episode draws its own number
  of start placements (env.py:384-387), so the envs finish at different calls: the start placements of the envs a
  call has just reset run inside that call while the others are held (`SRL_ACTION_HOLD`); this mode waits for every
  step (it needs the done flags on the host)."""

  def __init__(self, n_parallel=None, episode_length=15, n_objects=30, start_policy=None, min_episode_length=None,
               **kwargs):
    if n_objects < episode_length:
      raise ValueError("n_objects can't be less than episode_length. Got {} objects for {} steps long episodes.".format(
        n_objects, episode_length))                                            # env.py:375-378
    self._random_lengths = bool(min_episode_length and min_episode_length < episode_length)
    self._lower, self._upper = int(n_objects) - int(episode_length), int(n_objects) - int(min_episode_length or 0)
    super(StartedVecStackEnv, self).__init__(n_parallel=n_parallel, episode_length=n_objects, **kwargs)
    self._n_start_steps = int(n_objects) - int(episode_length)
    self._done_prev = None
    if start_policy is None:
      from stackrl_amd import baselines
      # lowest position with the object fully inside the goal: `height` values under the goal-overlap mask
      start_policy = baselines.Baseline('height', goal=True, minorder=0, threshold=1.0)
    elif not callable(start_policy):
      raise TypeError('Invalid type {} for argument start_policy. Must be callable.'.format(type(start_policy)))
    self._start_policy = start_policy
    self._since_reset = 0

  @property
  def n_start_steps(self):
    return self._n_start_steps

  def seed(self, seed):
    self._len_rng = np.random.RandomState(int(seed) % 2**32)   # the episode-length draws (env.py:387)
    return super(StartedVecStackEnv, self).seed(seed)

  def _start_random(self, step, fresh):
    """Start placements of the envs `fresh` (bool [B]) marks as just reset, each with its own count drawn from
    [n_objects - episode_length, n_objects - min_episode_length] (env.py:384-387); the others are held."""
    (om, oo), r, d = step
    counts = torch.from_numpy(self._len_rng.randint(self._lower, self._upper + 1, size=self._B)).to(self._device)
    todo = torch.where(fresh, counts, torch.zeros_like(counts))
    hold = torch.full((self._B,), _config.ACTION_HOLD, dtype=torch.int64, device=self._device)
    for j in range(int(todo.max())):
      a = torch.where(todo > j, self._start_policy((om, oo)).to(torch.int64), hold)
      (om, oo), _, _ = super(StartedVecStackEnv, self).step(a, block=True)   # a held env's observation does not change
    self.last_start_steps = todo
    keep = ~fresh
    return (om, oo), r * keep, d & keep

  def _start(self, step):
    for _ in range(self._n_start_steps):
      step = super(StartedVecStackEnv, self).step(self._start_policy(step[0]), block=True)
    self._since_reset = 0
    return (step[0], torch.zeros_like(step[1]), torch.zeros_like(step[2]))

  def reset(self, block=None):
    out = super(StartedVecStackEnv, self).reset(block=True)
    if self._random_lengths:
      out = self._start_random(out, torch.ones(self._B, dtype=torch.bool, device=self._device))
      self._done_prev = out[2]
    else:
      out = self._start(out)
    block = self._block if block is None else block
    return out if block else (lambda: out)

  def step(self, action, block=None):
    block = self._block if block is None else block
    if self._random_lengths:
      out = super(StartedVecStackEnv, self).step(action, block=True)
      fresh, self._done_prev = self._done_prev, out[2]
      if bool(fresh.any()):       # this call was the auto-reset of those envs (env.py:235-236)
        out = self._start_random(out, fresh)
      return out if block else (lambda: out)
    if self._since_reset == self.config.episode_length - self._n_start_steps:
      # every env is done: this call is the auto-reset (env.py:235-236), followed by the start placements
      out = self._start(super(StartedVecStackEnv, self).step(action, block=True))
      return out if block else (lambda: out)
    self._since_reset += 1
    return super(StartedVecStackEnv, self).step(action, block=block)


# constructor arguments of the reference's entry points, in signature order, with the registry's kwargs applied
# (env.py:28-51, :349-357, :444-449; envs/stack/__init__.py:4-27) — what `make(as_path=True)` names a directory after
_REF_ARGS = (('episode_length', 30), ('urdfs', '[5-9]?'), ('object_max_dimension', 0.125), ('use_gui', False),
             ('simulator', None), ('sim_time_step', 1 / 100.), ('gravity', 9.8), ('num_sim_steps', None),
             ('velocity_threshold', 0.01), ('smooth_placing', True), ('observer', None), ('observable_size_ratio', 4),
             ('resolution_factor', 5), ('max_z', 0.375), ('rewarder', None), ('goal_size_ratio', .25),
             ('reward_scale', 1.), ('reward_params', 2), ('flat_action', True), ('dtype', 'uint8'), ('seed', None))
_REF_ENTRY = {
  'Stack-v0': ('StackEnv', _REF_ARGS),
  # StartedStackEnv / TestStackEnv take their own arguments and pass the rest on as **kwargs, so only those appear
  # in the signature the reference inspects (utils.py:104-108)
  'Stack-v1': ('StartedStackEnv', (('episode_length', 15), ('min_episode_length', None), ('n_objects', 30),
                                   ('start_policy', None), ('flat_action', True), ('kwargs', None),
                                   ('urdfs', '[5-9]?'), ('reward_params', 2), ('dtype', 'uint8'))),
  'Stack-v2': ('TestStackEnv', (('ordering_freedom', False), ('orientation_freedom', 3), ('kwargs', None),
                                ('urdfs', '[5-9]?'), ('reward_params', 2), ('dtype', 'uint8'))),
}


def env_path(env='Stack-v0', **kwargs):
  """`make(as_path=True)` (utils.py:89-127): the directory name that identifies an env configuration — the entry
  point's class name, then one `<short name><value>` item per argument (all but `seed`): a name of several words is
  shortened to the first letter of the first and three letters of the last, a single word to four letters."""
  if env not in _REF_ENTRY:
    raise ValueError('Invalid env {}'.format(env))
  name, args = _REF_ENTRY[env]
  args = dict(args)
  args.update(kwargs)
  items = []
  for k, v in args.items():
    if k == 'seed':
      continue
    k = k.split('_')
    k = k[0][:1] + k[-1][:3] if len(k) > 1 else k[0][:4]
    items.append(k + str(v))
  return os.path.join(name, ','.join(items).replace(' ', '').replace("'", '').replace('"', ''))


def make(env='Stack-v0', n_parallel=None, block=None, seed=None, as_path=False, **kwargs):
  """`stackrl.envs.make` (utils.py:44-141): 'Stack-v0' (envs/stack/__init__.py:4-8), 'Stack-v1' (`StartedStackEnv`,
  env.py:348-441) and 'Stack-v2' (`TestStackEnv`, env.py:443-470, with its default `orientation_freedom=3`;
  `ordering_freedom=True` shows every rock of the episode and lets the action choose the next one)."""
  if as_path:
    return env_path(env, **kwargs)
  urdfs = kwargs.pop('urdfs', None)                      # env.py:92-103: which irregularity families the episode draws from
  if urdfs is not None:
    from stackrl_amd import assets
    kwargs['pool'] = (kwargs.get('pool') or assets.default_pool()).select(urdfs)
  if env == 'Stack-v2':
    kwargs.setdefault('orientation_freedom', 3)
  elif env == 'Stack-v1':
    return StartedVecStackEnv(n_parallel=n_parallel or 1, block=block, seed=seed, **kwargs)
  elif env != 'Stack-v0':
    raise ValueError("Invalid env {}: 'Stack-v0', 'Stack-v1' and 'Stack-v2' are implemented.".format(env))
  groups = kwargs.pop('groups', None)                    # build option: the batch as several handles (PipelinedVecStackEnv)
  if groups and (isinstance(groups, (tuple, list)) or int(groups) > 1):
    return PipelinedVecStackEnv(n_parallel=n_parallel or 1, groups=groups, block=block, seed=seed, **kwargs)
  return VecStackEnv(n_parallel=n_parallel or 1, block=block, seed=seed, **kwargs)


def make_curriculum(env='Stack-v0', n_parallel=None, block=None, curriculum=None, **kwargs):
  """`stackrl.envs.make_curriculum` (utils.py:143-182): a generator of `(env, goal)` tuples.  `curriculum` is a dict of
  equally long lists of `make` keyword arguments (e.g. `urdfs`) plus an optional list `goals` of goal returns."""
  curriculum = dict(curriculum or {})
  goals = curriculum.pop('goals', None)
  stages = [dict(zip(curriculum.keys(), values)) for values in zip(*curriculum.values())]
  if goals is not None and len(goals) != len(stages):
    raise ValueError("length of goals doesn't match number of environments")
  for i, cargs in enumerate(stages):
    instance = make(env, n_parallel=n_parallel, block=block, **cargs, **kwargs)
    yield instance, (None if goals is None else goals[i])
