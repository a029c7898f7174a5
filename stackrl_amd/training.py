"""The loop that calls the hot path: `Training.initialize` / `Training.run` (stackrl/train/training.py:233-296,
:338-380) restated as sequencing only — one vectorised env step and one minibatch update per iteration, the env
step enqueued non-blocking so that it overlaps `agent.train()` (training.py:359-368).  Logging, checkpoints, eval
and curriculum of the reference class are out of scope."""
import time

import torch


class Trainer(object):
  def __init__(self, env, agent):
    self._env, self._agent = env, agent
    self.collect_time = self.train_time = 0.0
    self.returns = torch.zeros(env.batch_size, device=agent.device)

  def initialize(self, num_steps=None, policy=None):
    """training.py:256-289: initial (random) collect; the last stored step is marked terminal."""
    env, agent = self._env, self._agent
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

  def run(self, max_num_iters):
    """training.py:335-380."""
    env, agent = self._env, self._agent
    step = env.reset()
    agent.acknowledge_reset()
    losses = []
    for _ in range(max_num_iters):
      t0 = time.perf_counter()
      if callable(step):
        step = step()
      self.returns += step[1]
      action = agent.collect(*step)
      step = env.step(action)            # non-blocking: the settle/render kernels overlap the update below
      t1 = time.perf_counter()
      loss, merr = agent.train()
      losses.append(loss)
      t2 = time.perf_counter()
      self.collect_time += t1 - t0
      self.train_time += t2 - t1
    if callable(step):
      step()
    return torch.stack(losses) if losses else torch.zeros(0)
