"""The loop that calls the hot path: `Training` (stackrl/train/training.py) restated — `initialize` (:233-296), `run`
(:298-423: one vectorised env step and one minibatch update per iteration, the env step enqueued non-blocking so that
it overlaps `agent.train()`, :359-368), `eval` (:398-452), `log_train` (:487-509), `save` (:454-465), `checkpoint`
(:467-485), with the reference's on-disk formats: `train.csv` (`Iter,Return,Loss,MeanError,CollectTime,TrainTime`),
`eval.csv` (`Iter,Return,Value,MeanValue,StdValue,MinValue,MaxValue`), `train.log`, `saved_weights/<iter>/weights`,
`checkpoint/`, `curriculum.csv` (`EndIter,Goal`).  Weights and checkpoints are torch files (the reference writes TF
checkpoints; TensorFlow is not part of this build).  Curriculum (:120-158, :521-575): `env` may be a generator of
`(env, goal)` tuples (`envs.make_curriculum`); when the training return passes `goal * (1 - epsilon)` the next
environment takes over."""
import os
import sys
import time
import traceback
import types
from datetime import datetime

import torch

from stackrl_amd import metrics


class Trainer(object):
  LOSS_HISTORY = 4096   # `run` returns the losses of at most this many last iterations

  def __init__(self, env, agent, eval_env=None, directory=None, log_interval=100, eval_interval=10000,
               checkpoint_interval=10000, eval_seed=None, train_reward_buffer_length=10, eval_reward_buffer_length=10,
               save_evaluated_policies=False, log_to_file=True, checkpoint_memory=True, goal_check_interval=1000):
    """
    Args (training.py:40-103):
      env, agent: the vectorised environment and the DQN agent.
      eval_env: environment for `eval()`; None disables evaluation.
      directory: where train.csv / eval.csv / train.log / saved_weights / checkpoint go; None disables all files.
      log_interval, eval_interval, checkpoint_interval: in agent iterations.
      eval_seed: seed of the evaluation env, reapplied at every evaluation (:404).
      train_reward_buffer_length, eval_reward_buffer_length: episodes averaged by the `Return` columns (:176-184).
      save_evaluated_policies: save the Q-net weights after every evaluation (:190-197, :386-387).
      checkpoint_memory: include the replay memory in checkpoints (the reference always does).
      goal_check_interval: with a curriculum, how often (iterations) the training return is checked against the goal.
    """
    # curriculum (training.py:120-158): generators of (env, goal)
    self._curriculum = self._eval_curriculum = None
    self._current_goal = None
    self._complete = False
    self._stop_when_complete = False
    self._goal_check_interval = None
    if isinstance(env, types.GeneratorType):
      self._curriculum = env
      env, self._current_goal = next(self._curriculum)
      if self._current_goal is None:
        raise ValueError('generator returned by env argument must yield tuples with env instance and goal')
      self._goal_check_interval = int(goal_check_interval)
    if isinstance(eval_env, types.GeneratorType):
      self._eval_curriculum = eval_env
      eval_env, _ = next(self._eval_curriculum)
    self._env, self._agent, self._eval_env = env, agent, eval_env
    self._directory = directory
    self._log_interval, self._eval_interval = int(log_interval), int(eval_interval)
    self._checkpoint_interval = int(checkpoint_interval)
    self._eval_seed = eval_seed
    self._save_weights = bool(save_evaluated_policies)
    self._checkpoint_memory = bool(checkpoint_memory)
    dev = agent.device
    self._reward = metrics.AverageReward(env.batch_size, length=train_reward_buffer_length, device=dev)
    self._eval_reward = metrics.AverageReward(eval_env.batch_size if eval_env is not None else 1,
                                              length=eval_reward_buffer_length, device=dev)
    self._loss = metrics.AverageMetric(length=self._log_interval, device=dev)
    self._mean_error = metrics.AverageMetric(length=self._log_interval, device=dev)
    self._collect_timer, self._train_timer = metrics.Timer(), metrics.Timer()
    self.collect_time = self.train_time = 0.0          # cumulative, for the benchmarks
    self._train_file = self._eval_file = self._log_file = self._ckpt_file = None
    # one writer per directory: with one process per GPU only rank 0 writes logs, weights and checkpoints (every rank
    # still restores from the shared checkpoint)
    import torch.distributed as dist
    self._rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    if directory is not None:
      os.makedirs(directory, exist_ok=True)
      self._train_file = os.path.join(directory, 'train.csv')
      self._eval_file = os.path.join(directory, 'eval.csv')
      self._log_file = os.path.join(directory, 'train.log') if log_to_file else None
      self._ckpt_file = os.path.join(directory, 'checkpoint', 'ckpt.pt')
    self._curriculum_file = os.path.join(directory, 'curriculum.csv') if directory is not None else None
    if self._curriculum is not None and self._curriculum_file is not None and os.path.isfile(self._curriculum_file):
      # skip the environments already solved in this train directory (training.py:131-153)
      with open(self._curriculum_file) as f:
        achieved = [float(line.split(',')[1]) for line in f.read().strip().split('\n')[1:] if line]
      for gdone in achieved:
        if gdone != self._current_goal:
          break
        if not self._advance():
          self._complete = True
          break
    self._last_checkpoint_iter = self._last_save_iter = None
    self._initialized = False
    self._reset_env = False

  # ------------------------------------------------------------------ small helpers
  @property
  def iterations(self):
    return self._agent.iterations

  @property
  def returns(self):
    """Average return of the last finished training episodes (`Return` column of train.csv)."""
    return self._reward.result

  def log(self, line):
    """training.py:577-588: timestamped line to train.log (or stdout)."""
    if self._rank != 0:
      return
    line = '{}: {}\n'.format(datetime.now(), line)
    if self._log_file is not None:
      with open(self._log_file, 'a') as f:
        f.write(line)
    elif self._directory is not None:
      sys.stdout.write(line)

  def log_exception(self):
    error = str(datetime.now()) + ': Exception.\n' + traceback.format_exc()
    if self._log_file is not None:
      with open(self._log_file, 'a') as f:
        f.write(error)
    else:
      sys.stderr.write(error)

  # ------------------------------------------------------------------ initialize (training.py:233-296)
  def initialize(self, num_steps=None, policy=None):
    """Restores the last checkpoint if there is one; otherwise the initial collect (random policy by default,
    training.py:256-263) whose last stored step is marked terminal (:284-289), then a first evaluation."""
    env, agent = self._env, self._agent
    if self._ckpt_file is not None and os.path.isfile(self._ckpt_file):
      self.log('Restoring checkpoint.')
      self.restore()
      # a checkpoint written with checkpoint_memory=False (an extension: the reference always saves the memory) leaves
      # the replay memory empty: sampling it would return unwritten slots, so the initial collect runs again
      if len(agent._replay_memory) >= agent._minibatch_size:
        self._initialized = True
        return
      self.log('Checkpoint holds no replay memory.')
    self.log('Collecting initial experience...')
    num_steps = num_steps or agent.replay_memory_size
    policy = policy or (lambda o: env.sample())
    step = env.reset()
    a = None
    for _ in range(num_steps - 1):
      if callable(step):
        step = step()
      a = policy(step[0])
      agent.observe(*step, a)
      step = env.step(a)
    o, r, _ = step() if callable(step) else step
    if a is None:
      a = policy(o)
    agent.observe(o, r, torch.ones(env.batch_size, dtype=torch.bool, device=r.device), a)
    self.log('Done.')
    if self._eval_env is not None and self._directory is not None:
      self.eval()
    self._initialized = True

  # ------------------------------------------------------------------ run (training.py:298-396)
  def run(self, max_num_iters=sys.maxsize, stop_when_complete=False):
    """stop_when_complete: with a curriculum, stop (StopIteration is swallowed) once the last goal is achieved; otherwise
    training continues on the last environment (training.py:305-309)."""
    self._stop_when_complete = bool(stop_when_complete)
    agent = self._agent
    if not self._initialized:
      self.initialize()
    env = self._env
    import collections
    losses = collections.deque(maxlen=self.LOSS_HISTORY)   # bounded: a real run lasts millions of iterations
    step = None
    failed = False
    try:
      step = env.reset()
      agent.acknowledge_reset()
      for _ in range(max_num_iters):
        t0 = time.perf_counter()
        with self._collect_timer:
          getattr(agent, 'train_begin', bool)()  # DQN(early_gradient=True): the gradient half of the update runs beside the collect step
          step = self.collect_step(env, step)   # non-blocking: the settle/render kernels overlap the update below
        t1 = time.perf_counter()
        with self._train_timer:
          loss, merr = agent.train()
          self._loss += loss
          self._mean_error += merr
        losses.append(loss)
        t2 = time.perf_counter()
        self.collect_time += t1 - t0
        self.train_time += t2 - t1
        iters = self.iterations
        if self._directory is not None:
          if iters % self._log_interval == 0:
            self.log_train()
          if self._eval_env is not None and iters % self._eval_interval == 0:
            self.eval()
            if self._save_weights:
              self.save()
          if iters % self._checkpoint_interval == 0:
            self.checkpoint()
        if self._goal_check_interval and iters % self._goal_check_interval == 0:
          self.check_goal()
        if self._reset_env:
          self._reset_env = False
          self._drain(env, step)
          env = self._env                 # the curriculum may have moved on
          step = env.reset()
          agent.acknowledge_reset()
    except StopIteration:
      self.log('Training goal achieved.')
    except Exception:
      failed = True
      self.log_exception()
      raise
    except BaseException:      # KeyboardInterrupt / SystemExit on this rank: its peers are still in the loop (see below)
      failed = True
      raise
    finally:
      self._drain(env, step)
      if self._directory is not None:
        # training.py:405-408 checkpoints in `finally`.  With several ranks a checkpoint is a collective (its barrier, the
        # shared file): a rank that left the loop by an exception must not enter it while its peers sit in the gradient
        # all-reduce, and what it holds may be half an update — the last periodic checkpoint stands instead.
        self.checkpoint(after_failure=failed)
    return torch.stack(list(losses)) if losses else torch.zeros(0)

  def collect_step(self, env, step):
    """The collect half of an iteration (training.py:352-357): the agent acts on the latest env step and stores it, the
    env starts carrying the actions out.  `step`: what the previous call (or `env.reset()`) returned; returns its
    successor.  An env held as groups of handles (`PipelinedVecStackEnv`) is served group by group — the policy call on
    a group waits for that group's step only, and the group's next step starts at once — with the random numbers of the
    whole batch drawn up front, so that the actions are those of one `agent.collect` over the batch."""
    agent = self._agent
    if getattr(env, 'groups', 1) > 1:
      draws = agent.policy_draws(env.batch_size)
      step, action = env.collect_step(
        lambda k, s, st: agent.policy(st[0], exploration=True, draws=tuple(d[s] for d in draws)))
      self._reward += step
      agent.observe(*step, action)
      return None                          # the env keeps the waits; `_drain` / the next call take them
    if callable(step):
      step = step()
    self._reward += step
    action = agent.collect(*step)
    return env.step(action)

  @staticmethod
  def _drain(env, step):
    if callable(step):
      step()
    getattr(env, 'drain', lambda: None)()

  # ------------------------------------------------------------------ eval (training.py:398-452)
  def eval(self):
    """Greedy policy on the evaluation env until `eval_reward_buffer_length` episodes finished; one eval.csv row."""
    env, agent = self._eval_env, self._agent
    self.log('Running evaluation...')
    self._eval_reward.reset(full=True)
    if self._eval_seed is not None:
      env.seed(self._eval_seed)
    step = env.reset()
    if callable(step):
      step = step()
    values = []
    while not self._eval_reward.full:
      a, value = agent.policy(step[0], values=True)
      step = env.step(a)
      if callable(step):
        step = step()
      self._eval_reward += step
      values.append(value)
    values = torch.stack(values)                                   # [steps, B, A]
    row = (self.iterations, float(self._eval_reward.result), float(values.amax(dim=-1).mean()), float(values.mean()),
           float(values.std(unbiased=False)), float(values.min()), float(values.max()))
    if self._eval_file is not None and self._rank == 0:
      header = '' if os.path.isfile(self._eval_file) else 'Iter,Return,Value,MeanValue,StdValue,MinValue,MaxValue\n'
      with open(self._eval_file, 'a') as f:
        f.write(header + '{},{},{},{},{},{},{}\n'.format(*row))
    self.log('Done.')
    return row

  # ------------------------------------------------------------------ curriculum (training.py:526-575)
  def _advance(self):
    """Next (env, goal) of the curriculum; False when it is exhausted."""
    try:
      new_env, self._current_goal = next(self._curriculum)
    except StopIteration:
      return False
    assert (tuple(new_env.observation_spec[0].shape), tuple(new_env.observation_spec[1].shape)) == \
           (tuple(self._env.observation_spec[0].shape), tuple(self._env.observation_spec[1].shape)), \
      'All envs in curriculum must have same observation and action specs.'
    old, self._env = self._env, new_env
    getattr(old, 'close', lambda: None)()
    if self._eval_curriculum is not None:
      new_eval, _ = next(self._eval_curriculum)
      old, self._eval_env = self._eval_env, new_eval
      getattr(old, 'close', lambda: None)()
    return True

  def check_goal(self):
    """training.py:526-546: goal reached when the training return exceeds goal * (1 - epsilon)."""
    # With several ranks the decision must be the same on all of them at the same iteration: a rank that advanced (or raised
    # StopIteration below) alone would leave its peers in the gradient all-reduce.  Every rank reaches this call at the same
    # iteration (goal_check_interval), so the training return is averaged over the ranks here — one scalar all-reduce —
    # and every rank compares the same number.
    ret = float(self._reward.result)
    if self._world() > 1:
      import torch.distributed as dist
      t = torch.tensor([ret], dtype=torch.float64, device=self._agent.device if dist.get_backend(getattr(self._agent, '_pg', None)) != 'gloo' else 'cpu')
      dist.all_reduce(t, group=getattr(self._agent, '_pg', None))
      ret = float(t.item()) / self._world()
    if not self._complete and ret > self._current_goal * (1 - self._agent.exploration):
      self.log('Goal reward achieved.')
      if self._curriculum_file is not None and self._rank == 0:
        header = '' if os.path.isfile(self._curriculum_file) else 'EndIter,Goal\n'
        with open(self._curriculum_file, 'a') as f:
          f.write(header + '{},{}\n'.format(self.iterations, self._current_goal))
      self.log('Updating environment...')
      if self._advance():
        self._reset_env = True            # triggers an environment reset in the training loop (:573-574)
        self.log('Done.')
      else:
        self._complete = True
    if self._complete and self._stop_when_complete:
      raise StopIteration('Training goal achieved.')

  # ------------------------------------------------------------------ logs, weights, checkpoints
  def log_train(self):
    """training.py:487-509: one train.csv row."""
    iters = self.iterations
    reward, loss, merr = float(self._reward.result), float(self._loss.result), float(self._mean_error.result)
    if self._train_file is not None and self._rank == 0:
      header = '' if os.path.isfile(self._train_file) else 'Iter,Return,Loss,MeanError,CollectTime,TrainTime\n'
      with open(self._train_file, 'a') as f:
        f.write(header + '{},{},{},{},{},{}\n'.format(iters, reward, loss, merr, self._collect_timer(), self._train_timer()))
    self.log('Iter {:8} Return {:<11.6} Loss {:<11.6}'.format(iters, reward, loss))

  def save(self):
    """training.py:454-465: the current Q-net weights under saved_weights/<iter>/weights."""
    iters = self.iterations
    if iters != self._last_save_iter and self._directory is not None and self._rank == 0:
      self.log("Saving Q network's weights...")
      path = os.path.join(self._directory, 'saved_weights', str(iters), 'weights')
      os.makedirs(os.path.dirname(path), exist_ok=True)
      self._agent.save_weights(path)
      self._last_save_iter = iters
      self.log('Done.')

  # Checkpoint files (training.py:467-485 keeps one `tf.train.Checkpoint`; there is one process per GPU here):
  #   checkpoint/ckpt.pt            what every replica shares — both nets, the optimiser slots, the iteration counter —
  #                                 written by rank 0 only (max_to_keep=1, never a half-written file)
  #   checkpoint/ckpt.rank<r>.pt    what belongs to rank r alone — its replay shard, the generators of its exploration
  #                                 and of its minibatch sampling, its training-return metric — written by rank r
  # so that after a resume the ranks do not hold copies of rank 0's replay memory and draw the same minibatch indices
  # and exploration numbers.  A barrier separates the writers from any reader.
  _SHARED_KEYS = ('q_net', 'target_q_net', 'optimizer', 'iterations')

  def _rank_file(self):
    return os.path.join(os.path.dirname(self._ckpt_file), 'ckpt.rank{}.pt'.format(self._rank))

  BARRIER_TIMEOUT_S = 600.0

  def _world(self):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
      return 1
    return dist.get_world_size(getattr(self._agent, '_pg', None))

  def _barrier(self):
    """Over the agent's process group (the one its gradient all-reduce uses); bounded where the backend can bound it."""
    import datetime
    import torch.distributed as dist
    if self._world() > 1:
      pg = getattr(self._agent, '_pg', None)
      if dist.get_backend(pg) == 'gloo':
        dist.monitored_barrier(group=pg, timeout=datetime.timedelta(seconds=self.BARRIER_TIMEOUT_S))
      else:
        dist.barrier(group=pg)

  def checkpoint(self, after_failure=False):
    """training.py:467-485: agent (nets, optimiser, counters, replay memory) + the training-return metric.
    after_failure: called while an exception is in flight — with one rank the reference's behaviour (save what there is),
    with several no file is touched and no collective entered."""
    iters = self.iterations
    if after_failure and self._world() > 1:
      self.log('Checkpoint skipped: this rank left the loop by an exception; the last periodic checkpoint stands.')
      return
    if iters != self._last_checkpoint_iter and self._ckpt_file is not None:
      self.log('Saving checkpoint...')
      os.makedirs(os.path.dirname(self._ckpt_file), exist_ok=True)
      d = self._agent.state_dict(memory=self._checkpoint_memory)
      local = {'agent': {k: v for k, v in d.items() if k not in self._SHARED_KEYS}, 'reward': self._reward.state_dict(),
               'iterations': iters}
      tmp = self._rank_file() + '.tmp'
      torch.save(local, tmp)
      os.replace(tmp, self._rank_file())
      if self._rank == 0:
        tmp = self._ckpt_file + '.tmp'
        torch.save({'agent': {k: d[k] for k in self._SHARED_KEYS}}, tmp)
        os.replace(tmp, self._ckpt_file)                           # max_to_keep=1, never a half-written file
      self._barrier()
      self._last_checkpoint_iter = iters
      self.log('Done.')

  def restore(self):
    self._barrier()
    d = torch.load(self._ckpt_file, map_location=self._agent.device, weights_only=False)
    self._agent.load_state_dict(d['agent'])
    mine = None
    if os.path.isfile(self._rank_file()):
      mine = torch.load(self._rank_file(), map_location=self._agent.device, weights_only=False)
      if mine.get('iterations') != self.iterations:                # a rank file of another checkpoint: not this run's state
        mine = None
    if mine is not None:
      self._agent.load_state_dict(mine['agent'])
      self._reward.load_state_dict(mine['reward'])
    elif 'gen' in d['agent'] and self._rank == 0:
      pass    # a file written before the per-rank split: the streams and the memory it carried are (rank 0's) own, loaded above
    else:
      # no state of this rank's own (e.g. resumed on more GPUs than the run was saved on): its generators restart from
      # rank-specific seeds and `initialize` collects its replay shard afresh
      self._agent.reseed(self.iterations * 1000003 + self._rank)
      if 'reward' in d:                                            # files written before the per-rank split
        self._reward.load_state_dict(d['reward'])
    self._last_checkpoint_iter = self.iterations
