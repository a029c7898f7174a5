"""Replay memory of the DQN update path: `ReplayMemory` (stackrl/agents/memory.py:9-361) on device tensors.

Partitioned circular buffer (one partition per env, memory.py:52-62), prioritised sampling without replacement
by Gumbel-top-k (memory.py:220-223), importance-sampling weights (memory.py:257-260), incremental min/max
logit tracking (memory.py:163-179, :278-316).  The reference pins its variables to the CPU (memory.py:49) and
ships every minibatch to the GPU; here everything stays in HBM (a B = 4,096, 64-slot memory is 8.9 GB).

Reference quirks, restated deliberately:
  * memory.py:239-242 computes the next index as `(i + n) % L + i // L` — for partition p > 0 that is
    `(k + n) % L + p`, a slot of partition 0, not of p (correct only for a single env).  The literal formula is the
    DEFAULT (results identical to the reference's come first); `reference_next_index=False` selects the intended
    "stay inside the partition" arithmetic `(k + n) % L + p * L` (tests pin both).
  * memory.py:169-172 / :306-309 mask with `logits * float(!isinf)`, i.e. -inf * 0 = NaN; the intent (minimum over
    the finite logits) is what is implemented.
  * `alpha * logits` with alpha = 0 gives NaN for unsampleable slots (memory.py:223); they are kept at -inf.
"""
import math

import torch


class ReplayMemory(object):
  def __init__(self, state_spec, max_length, alpha=None, beta=None, iters_counter=None, n_steps=None,
               epsilon=1e-3, seed=None, device=None, reference_next_index=True):
    """state_spec: sequence of (shape-with-batch, dtype), e.g. (((B,128,128,2), uint8), ((B,32,32,1), uint8)).
    alpha / beta: scalars or callables of the iteration count (memory.py:64-78)."""
    self.device = torch.device(device) if device is not None else torch.device('cpu')
    self._n_parts = int(state_spec[0][0][0])                       # memory.py:52
    max_length -= max_length % self._n_parts                       # memory.py:54
    self._max_length = max_length // self._n_parts                 # memory.py:56
    self._size = max_length
    self._offsets = torch.arange(self._n_parts, dtype=torch.int64, device=self.device) * self._max_length
    if (callable(alpha) or callable(beta)) and iters_counter is None:
      raise ValueError('iters_counter must be provided with a callable alpha/beta')   # memory.py:66-73
    self._alpha = alpha if callable(alpha) else float(alpha or 0.)
    self._beta = beta if callable(beta) else float(1. if beta is None else beta)
    self._iters_counter = iters_counter
    self._n_steps = int(n_steps or 1)
    self._n_range = torch.arange(1, self._n_steps + 1, dtype=torch.int64, device=self.device)
    assert self._max_length > self._n_steps                        # memory.py:86
    if epsilon <= 0:
      raise ValueError('epsilon must be greater than 0')           # memory.py:88-89
    self.epsilon = float(epsilon)
    self._gen = torch.Generator(device=self.device)
    if seed is not None:
      self._gen.manual_seed(int(seed))
    self._reference_next_index = bool(reference_next_index)
    self._states = [torch.zeros((max_length,) + tuple(shape[1:]), dtype=dtype, device=self.device)
                    for shape, dtype in state_spec]
    self._rewards = torch.zeros(max_length, dtype=torch.float32, device=self.device)
    self._terminal = torch.ones(max_length, dtype=torch.bool, device=self.device)      # memory.py:105-108
    self._actions = torch.zeros(max_length, dtype=torch.int64, device=self.device)
    self._logits = torch.full((max_length,), -math.inf, dtype=torch.float32, device=self.device)
    self._insert_index = 0
    # min / max logit trackers (memory.py:118-127) live on the device: no host round trip on the update path
    self._max_logit = torch.zeros((), dtype=torch.float32, device=self.device)
    self._max_logit_index = torch.zeros((), dtype=torch.int64, device=self.device)
    self._min_logit = torch.zeros((), dtype=torch.float32, device=self.device)
    self._min_logit_index = torch.zeros((), dtype=torch.int64, device=self.device)
    self.check = False      # True: raise like the reference's tf.debugging asserts (costs a device sync)
    # on a HIP device the scatter of `add`, the Gumbel top-k and the minibatch gather of `sample` are the hand-written
    # kernels of csrc/learner.hip (K7 / K8); they need two state tensors with rows of a multiple of 16 bytes
    self._fused = self.device.type == 'cuda' and len(self._states) == 2 and \
        all((s[0].numel() * s.element_size()) % 16 == 0 for s in self._states)
    self._ws = {}
    # schedule values as device scalars for a hipGraph-replayed update (refreshed by `refresh_schedules`)
    self._alpha_t = torch.zeros((), dtype=torch.float32, device=self.device)
    self._beta_t = torch.zeros((), dtype=torch.float32, device=self.device)
    self.tensor_schedules = False

  def __len__(self):
    return int(torch.isfinite(self._logits).sum())                 # memory.py:129-132

  # ------------------------------------------------------------------ checkpointing (the reference tracks these as
  # tf.Variables of the agent module, training.py:199-208)
  _STATE = ('_rewards', '_terminal', '_actions', '_logits', '_max_logit', '_max_logit_index', '_min_logit',
            '_min_logit_index')

  def state_dict(self):
    d = {k: getattr(self, k).clone() for k in self._STATE}
    d['states'] = [s.clone() for s in self._states]
    d['insert_index'] = int(self._insert_index)
    d['gen'] = self._gen.get_state()
    return d

  def load_state_dict(self, d):
    for k in self._STATE:
      getattr(self, k).copy_(d[k])
    for s, v in zip(self._states, d['states']):
      s.copy_(v)
    self._insert_index = int(d['insert_index'])
    self._gen.set_state(d['gen'].cpu())

  @property
  def max_length(self):
    return self._max_length

  @property
  def alpha(self):
    return float(self._alpha(self._iters_counter())) if callable(self._alpha) else self._alpha

  @property
  def beta(self):
    return float(self._beta(self._iters_counter())) if callable(self._beta) else self._beta

  # ------------------------------------------------------------------ add (memory.py:151-196)
  def _extrema(self):
    """((index, value) of the max logit, (index, value) of the min finite logit) over the whole memory.  On a HIP device one
    hand-written two-launch scan (csrc/learner.hip srl_logit_extrema): the framework's single-launch multi-block reductions
    are what returned garbage under the concurrent env step."""
    if self._fused and not self.check:
      from stackrl_amd import qops
      (mv, mi), (nv, ni) = qops.logit_extrema(self._logits, self._ws)
      return (mi, mv), (ni, nv)
    return self._argmax_all(), self._argmin_finite()

  def _argmax_all(self):
    v, idx = torch.max(self._logits, dim=0)         # (value, index) in one op: indexing with a device scalar would sync
    return idx, v

  def _argmin_finite(self):
    finite = torch.isfinite(self._logits)
    if self.check and not bool(finite.any()):
      raise FloatingPointError('No sampleable transition (failed to compute min logit)')   # memory.py:174-177
    masked = torch.where(finite, self._logits, torch.full_like(self._logits, math.inf))
    v, idx = torch.min(masked, dim=0)
    return idx, v

  def add(self, state, reward, terminal, action):
    L = self._max_length
    if self._fused:
      from stackrl_amd import qops
      qops.replay_scatter(tuple(u.to(v.dtype) for u, v in zip(state, self._states)), reward, terminal, action,
                          self._insert_index % L, L, self._states, self._rewards, self._terminal, self._actions, self._logits)
    else:
      idx = self._offsets + self._insert_index % L                 # memory.py:153
      for var, upd in zip(self._states, state):
        var.index_copy_(0, idx, upd.to(var.dtype))
      self._rewards.index_copy_(0, idx, reward.to(torch.float32))
      self._terminal.index_copy_(0, idx, terminal.to(torch.bool))
      self._actions.index_copy_(0, idx, action.to(torch.int64))
      self._logits.index_fill_(0, idx, -math.inf)                  # unsampleable until the next state exists
    if self._insert_index > 0:
      slot = self._insert_index % L
      # tf.reduce_any(index == indexes) (memory.py:164, :168): the tracked slot was just overwritten -> recompute
      hit_max = (self._max_logit_index % L) == slot
      hit_min = (self._min_logit_index % L) == slot
      (i, v), (i2, v2) = self._extrema()
      self._max_logit_index.copy_(torch.where(hit_max, i, self._max_logit_index))    # in place: the trackers keep
      self._max_logit.copy_(torch.where(hit_max, v, self._max_logit))                # their addresses (hipGraph replay)
      if self.check and bool(hit_min):
        self._argmin_finite()
      i, v = i2, v2
      self._min_logit_index.copy_(torch.where(hit_min, i, self._min_logit_index))
      self._min_logit.copy_(torch.where(hit_min, v, self._min_logit))
    # the transition n steps back becomes sampleable unless an episode boundary lies in between (memory.py:181-194)
    back = self._offsets[:, None] + ((self._insert_index - self._n_range) % L)[None, :]      # [B, n]
    boundary = self._terminal[back].any(dim=-1)
    val = torch.where(boundary, torch.full((), -math.inf, device=self.device), self._max_logit)
    self._logits.index_copy_(0, back[:, -1], val.to(torch.float32))
    self._insert_index += 1

  def set_terminal(self):
    """memory.py:199-203: mark the latest transition terminal (explicit reset after a non-terminal state)."""
    idx = self._offsets + (self._insert_index - 1) % self._max_length
    self._terminal.index_fill_(0, idx, True)

  # ------------------------------------------------------------------ sample (memory.py:206-263)
  def next_indexes(self, indexes, steps):
    L = self._max_length
    if self._reference_next_index:
      return (indexes + steps) % L + indexes // L                  # literal memory.py:239-242
    return (indexes % L + steps) % L + (indexes // L) * L

  def refresh_schedules(self):
    """Write the current alpha / beta (memory.py:133-149) into their device scalars; with `tensor_schedules` set,
    `sample` reads those instead of Python floats, so that a captured graph follows the schedules."""
    self._alpha_t.fill_(self.alpha); self._beta_t.fill_(self.beta)

  def sample(self, minibatch_size, get_weights=False):
    alpha = self._alpha_t if self.tensor_schedules else self.alpha
    u = torch.rand(self._logits.shape, generator=self._gen, device=self.device, dtype=torch.float32)
    if self._fused:
      return self._sample_fused(u, minibatch_size, get_weights)
    z = -torch.log(-torch.log(u))                                  # Gumbel-max trick, memory.py:220-222
    keys = torch.where(torch.isinf(self._logits), self._logits, alpha * self._logits) + z
    values, indexes = torch.topk(keys, minibatch_size)
    if self.check and not bool(torch.isfinite(values).all()):
      raise FloatingPointError('Not enough elements to sample')    # memory.py:227-230
    states = tuple(s[indexes] for s in self._states)
    actions = self._actions[indexes]
    nxt = self.next_indexes(indexes, self._n_steps)
    next_states = tuple(s[nxt] for s in self._states)
    terminal = self._terminal[nxt]
    if self._n_steps != 1:                                          # memory.py:251-254
      nxt = self.next_indexes(indexes[:, None], self._n_range[None, :])
    rewards = self._rewards[nxt]
    if get_weights:
      beta = self._beta_t if self.tensor_schedules else self.beta
      weights = torch.exp(beta * alpha * (self._min_logit - self._logits[indexes]))   # memory.py:257-260
      return indexes, weights, (states, actions, rewards, next_states, terminal)
    return states, actions, rewards, next_states, terminal

  def _sample_fused(self, u, minibatch_size, get_weights):
    """`sample` on the kernels of csrc/learner.hip: K7 (top-k of alpha logit + Gumbel(u)) and K8 (minibatch gather with
    the next-row arithmetic and the importance weights); the same draws `u` as the library formulation."""
    from stackrl_amd import qops
    if not self.tensor_schedules:
      self.refresh_schedules()
    indexes, keys = qops.gumbel_topk(self._logits, u, self._alpha_t, minibatch_size, self._ws)
    if self.check and not bool(torch.isfinite(keys).all()):
      raise FloatingPointError('Not enough elements to sample')    # memory.py:227-230
    if self._n_steps != 1:                                          # n-step rewards are gathered by the library path
      states = tuple(s[indexes] for s in self._states)
      nxt = self.next_indexes(indexes, self._n_steps)
      batch = (states, self._actions[indexes], self._rewards[self.next_indexes(indexes[:, None], self._n_range[None, :])],
               tuple(s[nxt] for s in self._states), self._terminal[nxt])
      weights = torch.exp(self._beta_t * self._alpha_t * (self._min_logit - self._logits[indexes])) if get_weights else None
    else:
      batch, weights = qops.replay_gather(indexes, self._max_length, 1, self._reference_next_index, self._states, self._rewards,
                                          self._terminal, self._actions, self._logits,
                                          *((self._alpha_t, self._beta_t, self._min_logit) if get_weights else ()))
    if get_weights:
      return indexes, weights, batch
    return batch

  # ------------------------------------------------------------------ update_priorities (memory.py:266-316)
  def update_priorities(self, indexes, deltas, logits=None):
    if logits is None:
      logits = torch.log(deltas.to(torch.float32) + self.epsilon)  # memory.py:272
    self._logits.index_copy_(0, indexes, logits)
    max_logit, amax = torch.max(logits, dim=0); min_logit, amin = torch.min(logits, dim=0)
    imax = indexes.gather(0, amax.view(1))[0]; imin = indexes.gather(0, amin.view(1))[0]
    hit_max = (indexes == self._max_logit_index).any()
    hit_min = (indexes == self._min_logit_index).any()
    # memory.py:282-292: a larger maximum replaces the tracker; else if the tracked slot was rewritten, recompute
    (ri, rv), (ri2, rv2) = self._extrema()
    ge = max_logit >= self._max_logit
    self._max_logit_index.copy_(torch.where(ge, imax, torch.where(hit_max, ri, self._max_logit_index)))
    self._max_logit.copy_(torch.where(ge, max_logit, torch.where(hit_max, rv, self._max_logit)))
    ri, rv = ri2, rv2                                              # memory.py:298-316
    le = min_logit <= self._min_logit
    self._min_logit_index.copy_(torch.where(le, imin, torch.where(hit_min, ri, self._min_logit_index)))
    self._min_logit.copy_(torch.where(le, min_logit, torch.where(hit_min, rv, self._min_logit)))
