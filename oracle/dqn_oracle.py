"""CPU restatement (numpy, float64 / pure-Python loops) of the learner rows.  TEST INFRASTRUCTURE ONLY.

  * `RefMemory`     stackrl/agents/memory.py:151-196 (add), :199-203 (set_terminal), :239-260 (next index, rewards,
                    importance weights), :266-316 (update_priorities) — list-based, one slot at a time.
  * `dqn_targets`   stackrl/agents/dqn.py:408-469 (one-hot gather, Double-DQN, terminal masking, Huber, IS weights)
  * `xcorr`         stackrl/nets/layers.py:21-38 as explicit loops
Parity pinning: no reference tests or golden vectors exist for these rows and TensorFlow is not
installed, so this oracle is a line-by-line restatement ("parity unpinned" against TF itself; closed forms only).
"""
import math

import numpy as np

NEG_INF = -math.inf


class RefMemory(object):
  def __init__(self, n_parts, max_length, n_steps=1, epsilon=1e-3, literal_next_index=True):   # memory.py:239-242 as written
    max_length -= max_length % n_parts                  # memory.py:54
    self.P, self.L, self.n = n_parts, max_length // n_parts, n_steps
    N = max_length
    self.rewards = [0.0] * N
    self.terminal = [True] * N                          # memory.py:105-108
    self.actions = [0] * N
    self.logits = [NEG_INF] * N
    self.states = [None] * N
    self.insert = 0
    self.max_logit, self.max_idx, self.min_logit, self.min_idx = 0.0, 0, 0.0, 0
    self.eps = epsilon
    self.literal = literal_next_index

  def _argmax(self):
    best = 0
    for i, v in enumerate(self.logits):
      if v > self.logits[best]:
        best = i
    return best

  def _argmin_finite(self):
    best = None
    for i, v in enumerate(self.logits):
      if v != NEG_INF and (best is None or v < self.logits[best]):
        best = i
    assert best is not None, 'No sampleable transition (failed to compute min logit)'
    return best

  def add(self, state, reward, terminal, action):
    idx = [p * self.L + self.insert % self.L for p in range(self.P)]          # memory.py:153
    for p, i in enumerate(idx):
      self.states[i] = state[p]; self.rewards[i] = float(reward[p])
      self.terminal[i] = bool(terminal[p]); self.actions[i] = int(action[p])
      self.logits[i] = NEG_INF                                                # memory.py:161
    if self.max_idx in idx and self.insert > 0:                               # memory.py:164-167
      self.max_idx = self._argmax(); self.max_logit = self.logits[self.max_idx]
    if self.min_idx in idx and self.insert > 0:                               # memory.py:168-179
      self.min_idx = self._argmin_finite(); self.min_logit = self.logits[self.min_idx]
    for p in range(self.P):                                                   # memory.py:183-194
      back = [p * self.L + (self.insert - k) % self.L for k in range(1, self.n + 1)]
      boundary = any(self.terminal[i] for i in back)
      self.logits[back[-1]] = NEG_INF if boundary else self.max_logit
    self.insert += 1

  def set_terminal(self):
    for p in range(self.P):
      self.terminal[p * self.L + (self.insert - 1) % self.L] = True          # memory.py:202-203

  def next_index(self, i, steps):
    if self.literal:
      return (i + steps) % self.L + i // self.L                               # memory.py:239-242 as written
    return (i % self.L + steps) % self.L + (i // self.L) * self.L

  def transition(self, i):
    """What `sample` returns for index i (memory.py:232-256)."""
    nxt = self.next_index(i, self.n)
    rew = self.rewards[nxt] if self.n == 1 else [self.rewards[self.next_index(i, k)] for k in range(1, self.n + 1)]
    return self.states[i], self.actions[i], rew, self.states[nxt], self.terminal[nxt]

  def weight(self, i, alpha, beta):
    return math.exp(beta * alpha * (self.min_logit - self.logits[i]))         # memory.py:257-260

  def update_priorities(self, indexes, deltas):
    logits = [math.log(float(np.float32(d) + np.float32(self.eps))) for d in deltas]   # float32 add like the tensors
    logits = [float(np.float32(np.log(np.float32(d) + np.float32(self.eps)))) for d in deltas]
    for i, l in zip(indexes, logits):
      self.logits[i] = l
    amax = int(np.argmax(logits)); amin = int(np.argmin(logits))
    if logits[amax] >= self.max_logit:                                        # memory.py:282-292
      self.max_idx, self.max_logit = indexes[amax], logits[amax]
    elif self.max_idx in indexes:
      self.max_idx = self._argmax(); self.max_logit = self.logits[self.max_idx]
    if logits[amin] <= self.min_logit:                                        # memory.py:298-316
      self.min_idx, self.min_logit = indexes[amin], logits[amin]
    elif self.min_idx in indexes:
      self.min_idx = self._argmin_finite(); self.min_logit = self.logits[self.min_idx]


def dqn_targets(q, q_next_online, q_next_target, actions, rewards, terminal, gamma, double=True, huber_delta=1.0,
                weights=None):
  """dqn.py:408-469 in float64: returns (loss, mean td, |td|)."""
  q = np.asarray(q, np.float64); qo = np.asarray(q_next_online, np.float64); qt = np.asarray(q_next_target, np.float64)
  n = q.shape[0]
  qa = q[np.arange(n), actions]                                   # one_hot . reduce_sum, dqn.py:410-417
  if double:
    tq = qt[np.arange(n), np.argmax(qo, axis=-1)]                 # dqn.py:424-431
  else:
    tq = qt.max(axis=-1)
  y = np.asarray(rewards, np.float64) + np.where(terminal, 0.0, gamma * tq)   # dqn.py:450-454
  td = qa - y
  mtd = td.mean()
  td = np.abs(td)
  if huber_delta is not None:
    quad = np.minimum(td, huber_delta); lin = td - quad
    loss = 0.5 * quad ** 2 + huber_delta * lin                    # dqn.py:461-464
  else:
    loss = 0.5 * td ** 2
  if weights is not None:
    loss = loss * np.asarray(weights, np.float64)
  return loss.mean(), mtd, td


def xcorr(x, w):
  """layers.py:21-38: x [B,C,H,W], w [B,C,h,w] -> [B,1,H-h+1,W-w+1], float64 loops."""
  x = np.asarray(x, np.float64); w = np.asarray(w, np.float64)
  B, C, H, W = x.shape; h, ww = w.shape[-2:]
  out = np.zeros((B, 1, H - h + 1, W - ww + 1))
  for b in range(B):
    for u in range(H - h + 1):
      for v in range(W - ww + 1):
        out[b, 0, u, v] = (x[b, :, u:u + h, v:v + ww] * w[b]).sum()
  return out
