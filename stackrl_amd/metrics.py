"""Training metrics of the reference (stackrl/metrics.py): `Timer` (:5-87), `AverageMetric` (:89-136) and
`AverageReward` (:138-173), on torch tensors.  The buffers live on the metric's device and are updated without a
host round trip (the reference loops over `tf.where(terminal)` in Python); reading `.result` is where a sync
happens, i.e. at the log interval."""
import time

import torch


class Timer(object):
  """Times `with` blocks.  `timer()` gives the mean duration of the blocks finished since the last reset (None when
  there is none) and, unless `reset=False`, starts a new averaging window — the `CollectTime` / `TrainTime` columns of
  train.csv (training.py:495-505).  Only the surface of the reference's `metrics.Timer` is kept (context manager + call);
  durations are measured per block with a monotonic clock."""

  def __init__(self, clock=None):
    if isinstance(clock, str):
      clock = getattr(time, clock)
    if clock is not None and not callable(clock):
      raise TypeError('clock must be None, the name of a function of the time module, or a callable')
    self._clock = clock or time.perf_counter
    self._t0 = None
    self.total, self.n = 0.0, 0

  def reset(self):
    self.total, self.n = 0.0, 0

  def __enter__(self):
    self._t0 = self._clock()
    return self

  def __exit__(self, *exc):
    self.total += self._clock() - self._t0
    self.n += 1
    self._t0 = None
    return False

  def __call__(self, reset=True):
    if self.n == 0:
      return None
    mean = self.total / self.n
    if reset:
      self.reset()
    return mean


class AverageMetric(object):
  """Mean of the last `length` added values (of all of them until `length` were added), metrics.py:89-136."""

  def __init__(self, length=100, dtype=torch.float32, device=None):
    self._length = int(length)
    self._index = torch.zeros((), dtype=torch.int64, device=device)
    self._buf = torch.zeros(self._length + 1, dtype=dtype, device=device)   # + one scratch slot for masked-out writes

  @property
  def _values(self):
    return self._buf[:self._length]

  @property
  def result(self):
    n = torch.clamp(self._index, max=self._length).to(self._values.dtype)
    return self._values.sum() / n          # values beyond `index` are zero; 0/0 = nan before the first add, as in tf

  @property
  def full(self):
    return bool(self._index >= self._length)

  def add(self, value):
    value = torch.as_tensor(value, dtype=self._buf.dtype, device=self._buf.device).detach().reshape(1)
    self._buf.index_copy_(0, (self._index % self._length).reshape(1), value)   # no host round trip
    self._index += 1

  def reset(self):
    self._buf.zero_()
    self._index.zero_()

  def __call__(self, *args, **kwargs):
    return self.add(*args, **kwargs)

  def __iadd__(self, other):
    self.add(other)
    return self

  def __lt__(self, other): return bool(self.result < other)
  def __le__(self, other): return bool(self.result <= other)
  def __ge__(self, other): return bool(self.result >= other)
  def __gt__(self, other): return bool(self.result > other)

  def state_dict(self):
    return {'index': self._index.clone(), 'values': self._values.clone()}

  def load_state_dict(self, d):
    self._index.copy_(d['index']); self._values.copy_(d['values'])   # `_values` is a view of the buffer


class AverageReward(AverageMetric):
  """Average return of the last `length` finished episodes of a batch of envs, metrics.py:138-173.  `add(step)` takes
  a collection whose last two elements are the rewards and terminal flags of one vectorised step; episodes that end
  in the same step enter the buffer in env order, exactly like the reference's loop."""

  def __init__(self, batch_size, length=100, dtype=torch.float32, device=None):
    super(AverageReward, self).__init__(length=length, dtype=dtype, device=device)
    self._episode_reward = torch.zeros(int(batch_size), dtype=dtype, device=device)

  def add(self, step):
    reward, terminal = step[-2], step[-1]
    dev = self._values.device
    reward = reward.to(device=dev, dtype=self._values.dtype); terminal = terminal.to(device=dev, dtype=torch.bool)
    self._episode_reward += reward
    rank = torch.cumsum(terminal.to(torch.int64), 0) - 1        # order of the finished episodes in this step
    k = terminal.sum()
    keep = terminal & (rank >= k - self._length)                # more than `length` at once: the last ones survive
    pos = torch.where(keep, (self._index + rank) % self._length, torch.full_like(rank, self._length))
    self._buf.index_copy_(0, pos, self._episode_reward)         # masked-out envs land in the scratch slot
    self._buf[self._length] = 0
    self._index += k
    self._episode_reward = torch.where(terminal, torch.zeros_like(self._episode_reward), self._episode_reward)

  def reset(self, full=False):
    super(AverageReward, self).reset()
    if full:
      self._episode_reward.zero_()

  def state_dict(self):
    d = super(AverageReward, self).state_dict()
    d['episode_reward'] = self._episode_reward.clone()
    return d

  def load_state_dict(self, d):
    super(AverageReward, self).load_state_dict(d)
    self._episode_reward.copy_(d['episode_reward'])
