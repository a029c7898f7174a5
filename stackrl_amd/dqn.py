"""DQN agent of the rollout/update path: `DQN` (stackrl/agents/dqn.py:19-486) on PyTorch-ROCm.

Same surface as the reference: `collect` (:391-395), `observe` (:387-389), `train` (:397-486), `policy` (:330-375),
`acknowledge_reset` (:381-385), `iterations`, `epsilon`, `exploration`, `replay_memory_size`.  Epsilon-greedy or
Boltzmann (Gumbel-max) exploration, Huber TD loss, Double-DQN, n-step returns, prioritised replay with
importance-sampling weights, hard target sync every `target_update_period` iterations.

Multi-GPU (absent in the reference): one process per GPU, env shard + replay shard per rank, and ONE collective
per update — an all-reduce (RCCL over xGMI) of the flat gradient bucket (~2.13 M fp32 = 8.5 MB), averaged.
All gradients are views of one contiguous buffer, so the collective needs no packing copies.
"""
import math

import torch
import torch.distributed as dist

from stackrl_amd.memory import ReplayMemory


# Stream captures run in thread-local error mode: with a process group alive (RCCL's watchdog thread polls its events)
# another thread's HIP calls must not invalidate a capture in progress on this one
CAPTURE_MODE = 'thread_local'


class PolynomialDecay(object):
  """keras.optimizers.schedules.PolynomialDecay (config.gin:73-81), cycle=False."""

  def __init__(self, initial_learning_rate, decay_steps, end_learning_rate=0.0001, power=1.0):
    self.initial, self.steps, self.end, self.power = float(initial_learning_rate), int(decay_steps), float(end_learning_rate), float(power)

  def __call__(self, step):
    s = min(int(step), self.steps)
    return (self.initial - self.end) * (1 - s / self.steps) ** self.power + self.end


EXPLORATION_MODES = ['epsilon-greedy', 'boltzmann']     # dqn.py:23-28


class KerasAdam(object):
  """`keras.optimizers.Adam` as the reference applies it (dqn.py:473; config.gin:90-93) over ONE flat fp32 bucket:
  m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); p -= lr_t m / (sqrt(v) + eps), lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)
  (epsilon outside the bias correction, unlike torch.optim.Adam).  The parameters are re-pointed to views of the
  bucket, so a step is one launch of `srl_adam_step` (csrc/learner.hip) on a HIP device — capturable, the step counter
  and its powers live in device memory — and the same formula in torch on the CPU (host-logic tests)."""

  ALIGN = 64   # every parameter starts on a 256-byte boundary of the bucket (library kernels load weights in vectors)

  @classmethod
  def layout(cls, params):
    """Offsets of the parameters in a flat bucket and the bucket's length (elements)."""
    offs, o = [], 0
    for p in params:
      offs.append(o)
      o += (p.numel() + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN
    return offs, o

  def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-7):
    self.params = list(params)
    dev = self.params[0].device
    self.lr, (self.b1, self.b2), self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
    offs, n = self.layout(self.params)
    self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
    with torch.no_grad():
      for p, o in zip(self.params, offs):
        self.flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
        p.data = self.flat[o:o + p.numel()].view_as(p)
    self.m = torch.zeros_like(self.flat)
    self.v = torch.zeros_like(self.flat)
    self.state = torch.tensor([0.0, 1.0, 1.0, 0.0], dtype=torch.float32, device=dev)   # t, b1^t, b2^t, lr_t

  @torch.no_grad()
  def step(self, flat_grad):
    if self.flat.is_cuda:
      from stackrl_amd import qops
      qops.adam_step(self.flat, flat_grad, self.m, self.v, self.state, self.lr, self.b1, self.b2, self.eps)
      return
    st = self.state
    st[0] += 1.0; st[1] *= self.b1; st[2] *= self.b2
    st[3] = self.lr * torch.sqrt(1.0 - st[2]) / (1.0 - st[1])
    self.m += (flat_grad - self.m) * (1.0 - self.b1)
    self.v += (flat_grad * flat_grad - self.v) * (1.0 - self.b2)
    self.flat -= (st[3] * self.m) / (torch.sqrt(self.v) + self.eps)

  def state_dict(self):
    return {'m': self.m.clone(), 'v': self.v.clone(), 'state': self.state.clone()}

  def load_state_dict(self, d):
    self.m.copy_(d['m']); self.v.copy_(d['v']); self.state.copy_(d['state'])


class GraphedEval(object):
  """`net(inputs)` without grad, replayed from a hipGraph (torch.cuda.CUDAGraph) for a fixed input signature.
  The target evaluations of the update are ~200 launches of microsecond kernels each; as a graph each is one launch.
  The net's parameters are read in place, so optimiser steps and target syncs need no re-capture."""

  def __init__(self, net):
    self.net = net
    self.graph = None
    self.sig = None

  def _capture(self, inputs):
    self.static_in = tuple(torch.empty_like(t) for t in inputs)
    for s, t in zip(self.static_in, inputs):
      s.copy_(t)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():       # warm-up: library solver search, lazy initialisations
      for _ in range(2):
        self.net(self.static_in)
    torch.cuda.current_stream().wait_stream(side)
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE), torch.no_grad():
      self.static_out = self.net(self.static_in)

  def __call__(self, inputs):
    sig = tuple((tuple(t.shape), t.dtype) for t in inputs)
    if sig != self.sig:
      self._capture(inputs)
      self.sig = sig
    for s, t in zip(self.static_in, inputs):
      s.copy_(t)
    self.graph.replay()
    return self.static_out


class DQN(object):
  def __init__(self, q_net, optimizer=None, learning_rate=None, huber_delta=1., minibatch_size=32,
               replay_memory_size=100000, prefetch=None, target_update_period=10000, reward_scale=None,
               discount_factor=.99, collect_batch_size=None, exploration_mode=None, exploration=None,
               prioritization=None, priority_bias_compensation=None, double=False, n_step=None, seed=None,
               device=None, process_group=None, policy_op=None, reference_next_index=True,
               adam_betas=(0.9, 0.999), xcorr=None, graphs=False, hand_convs=None, early_gradient=False):
    if not isinstance(q_net, torch.nn.Module):
      raise TypeError('Invalid type {} for argument q_net. Must be a torch Module.'.format(type(q_net)))   # dqn.py:122-125
    self.device = torch.device(device) if device is not None else next(q_net.parameters()).device
    self._q_net = q_net.to(self.device)
    # xcorr: how `layers.correlation` (layers.py:21-38) is evaluated and differentiated in the update.  None keeps the
    # net's own (library grouped convolution, fp32); 'bf16x3' / 'bf16' use the MFMA kernels of csrc/xcorr_mfma.hip
    # (hi/lo-split fp32-class accuracy / operands rounded to bf16) on a HIP device.
    if xcorr is not None:
      if xcorr not in ('bf16x3', 'bf16'):
        raise ValueError("Invalid value {} for argument xcorr. Must be None, 'bf16x3' or 'bf16'.".format(xcorr))
      if self.device.type == 'cuda' and hasattr(self._q_net, 'correlation'):
        from stackrl_amd import qops
        self._q_net.correlation = qops.correlation(qops.BF16X3 if xcorr == 'bf16x3' else qops.BF16)
    # on a HIP device the bias add + ReLU of every convolution (forward, backward and the bias-gradient reduction) are
    # the fused passes of csrc/epilogue.hip
    if self.device.type == 'cuda' and hasattr(self._q_net, 'set_fused_epilogues'):
      self._q_net.set_fused_epilogues(True)
    self._pg = process_group
    self._world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
    if self._world > 1:
      # replicas must start from the same weights whatever seed each rank built its net with: rank 0's are broadcast
      # (the gradient all-reduce keeps them equal from then on)
      src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
      with torch.no_grad():
        for t in list(self._q_net.parameters()) + list(self._q_net.buffers()):
          dist.broadcast(t, src=src, group=process_group)
    import copy
    self._target_q_net = copy.deepcopy(self._q_net)                 # clone + set_weights, dqn.py:116-117
    for p in self._target_q_net.parameters():
      p.requires_grad_(False)
    # one flat gradient bucket; every p.grad is a view into it
    params = [p for p in self._q_net.parameters() if p.requires_grad]
    offs, n = KerasAdam.layout(params)        # the same (256-byte aligned) layout as the optimiser's parameter bucket
    self._flat_grad = torch.zeros(n, dtype=torch.float32, device=self.device)
    for p, o in zip(params, offs):
      p.grad = self._flat_grad[o:o + p.numel()].view_as(p)
    self._params = params
    if optimizer is None:                                            # dqn.py:127-130: Adam with Keras' defaults
      optimizer = KerasAdam(params, lr=learning_rate or 0.00025, betas=adam_betas, eps=1e-7)
    elif callable(optimizer) and not isinstance(optimizer, torch.optim.Optimizer):
      optimizer = optimizer(params, lr=learning_rate or 0.00025)
    elif not isinstance(optimizer, (torch.optim.Optimizer, KerasAdam)):
      raise TypeError('Invalid type {} for argument optimizer.'.format(type(optimizer)))
    self._optimizer = optimizer
    self._iterations = 0
    # exploration (dqn.py:140-187)
    if exploration_mode is None:
      self._exploration_mode = EXPLORATION_MODES[0]
    elif isinstance(exploration_mode, int):
      self._exploration_mode = EXPLORATION_MODES[exploration_mode]
    elif isinstance(exploration_mode, str):
      if exploration_mode.lower() not in EXPLORATION_MODES:
        raise ValueError('Invalid value {} for argument exploration_mode. Must be in {}.'.format(exploration_mode, EXPLORATION_MODES))
      self._exploration_mode = exploration_mode.lower()
    else:
      raise TypeError('Invalid type {} for argument exploration_mode. Must be int or str.'.format(type(exploration_mode)))
    if exploration is None:
      exploration = 0.1 if self._exploration_mode == 'epsilon-greedy' else 1.
    dummy = exploration(0) if callable(exploration) else exploration
    if self._exploration_mode == 'epsilon-greedy' and (dummy < 0 or dummy > 1):
      raise ValueError('Invalid value {} for argument exploration. Must be in [0,1].'.format(exploration))
    if self._exploration_mode == 'boltzmann' and dummy <= 0:
      raise ValueError('Invalid value {} for argument exploration. Must be greater than 0.'.format(exploration))
    self._exploration = exploration
    self._huber = huber_delta is not None                            # dqn.py:189-192
    self._huber_delta = float(huber_delta) if self._huber else None
    self._target_update_period = target_update_period or 10000
    n_step = n_step or 1                                             # dqn.py:195-211
    self._n_step = n_step > 1
    if self._n_step:
      self._gamma_r = torch.tensor([discount_factor ** i for i in range(n_step)], dtype=torch.float32, device=self.device)
      self._gamma = float(discount_factor ** n_step)
    else:
      self._gamma = float(discount_factor)
    self._reward_scale = float(reward_scale) if reward_scale else None
    self._minibatch_size = int(minibatch_size)
    collect_batch_size = collect_batch_size or 1
    self._n_actions = int(q_net.n_actions)
    prioritization = prioritization or 0.                            # dqn.py:229-235
    self._prioritized = prioritization != 0.
    self._bias_compensation = False
    if self._prioritized:
      if priority_bias_compensation is None:
        priority_bias_compensation = 1.
      self._bias_compensation = callable(priority_bias_compensation) or priority_bias_compensation != 0.
    (H, W), (h, w) = getattr(q_net, 'in_hw', ((128, 128), (32, 32)))
    state_spec = (((collect_batch_size, H, W, 2), torch.uint8), ((collect_batch_size, h, w, 1), torch.uint8))
    self._replay_memory = ReplayMemory(state_spec, replay_memory_size, alpha=prioritization,
                                       beta=priority_bias_compensation, iters_counter=lambda: self._iterations,
                                       n_steps=n_step, seed=seed, device=self.device,
                                       reference_next_index=reference_next_index)
    self._double = double
    # prefetch: how many minibatches are sampled ahead of the update that uses them (dqn.py:247-252; None / 0 = none)
    self._prefetch = int(prefetch or 0)
    self._fifo = None
    self._last_sample_indexes = None
    self._gen = torch.Generator(device=self.device)
    if seed is not None:
      self._gen.manual_seed(int(seed) + 1)
    self._policy_op = policy_op     # optional fused rollout head (stackrl_amd.qops.FusedPolicy)
    self.policy_timer = None        # optional object with start() / stop() called around every policy evaluation (bench.py)
    # graphs: replay the no-grad target evaluations of the update from hipGraphs (HIP device only)
    self._graphs = bool(graphs) and self.device.type == 'cuda'
    self._g_target = GraphedEval(self._target_q_net) if self._graphs else None
    self._g_online = GraphedEval(self._q_net) if self._graphs else None
    self._train_graph = None
    self._graph_calls = 0
    # early_gradient: `train_begin()` starts the gradient half of the next update — target evaluation, forward, loss,
    # backward, on the minibatch at the head of the prefetch FIFO, which was drawn `prefetch` updates ago and does not depend
    # on the collect step of this iteration — on a stream of its own, beside the collect step; `train()` then completes the
    # update (new minibatch into the FIFO, all-reduce, optimiser step, priorities).  Same operations on the same operands in
    # the same order per tensor as the serial update: identical results (tests/test_learner_gpu.py).  Needs prefetch >= 1.
    self._early = bool(early_gradient) and self.device.type == 'cuda' and self._prefetch >= 1
    self._upd_stream = torch.cuda.Stream(device=self.device) if self._early else None
    self._pending = None                    # outputs of a gradient half started by train_begin()
    self._grad_graph = self._tail_graph = None
    # the loss and its gradient as one hand-written kernel (csrc/learner.hip) on a HIP device
    self._fused = self.device.type == 'cuda'
    self._ws = {}
    # hand_convs: every convolution of the update (forward with saved activations, data gradient, weight gradient) on the
    # kernels of csrc/train_conv.hip instead of the library's (stackrl_amd/qtrain.py; None = on a HIP device whenever the
    # net and the optimiser allow it: the config.gin network over the flat Keras-Adam bucket)
    self._hand = self._hand_t = None
    if hand_convs is None:
      hand_convs = self._fused and isinstance(self._optimizer, KerasAdam) and self._hand_supported()
    if hand_convs:
      from stackrl_amd import qtrain
      self._hand = qtrain.HandNet(self._q_net, flat=self._optimizer.flat)
      self._hand_t = qtrain.HandNet(self._target_q_net)
      self._hand_t.refresh()

  def _hand_supported(self):
    """`qtrain.HandNet` covers the config.gin network: a `DeepQSiamFCN` with the dueling head on average-pooled features,
    two position layers, and the observation sizes the cross-correlation kernels are built for."""
    from stackrl_amd import nets, qops
    n = self._q_net
    if not isinstance(n, nets.DeepQSiamFCN) or not n.dueling or not n.dueling_avg_pool or len(n.pos) != 5:
      return False
    (H, W), (h, w) = n.in_hw
    chans = [m.out_channels for m in n.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and m.out_channels > 1]
    return H == W and h == w and (H, h) in qops.MFMA_SHAPES and all(c % 16 == 0 and c <= 256 for c in chans)

  def __call__(self, state, reward, terminal, action=None):
    return self.collect(state, reward, terminal) if action is None else self.observe(state, reward, terminal, action)

  # ------------------------------------------------------------------ properties (dqn.py:307-328)
  @property
  def iterations(self):
    return self._iterations

  @property
  def replay_memory_size(self):
    return self._replay_memory.max_length

  @property
  def exploration(self):
    return float(self._exploration(self._iterations)) if callable(self._exploration) else float(self._exploration)

  @property
  def epsilon(self):
    if self._exploration_mode == 'epsilon-greedy':
      return self.exploration
    return math.exp(-1 / self.exploration)

  @property
  def q_net(self):
    return self._q_net

  # ------------------------------------------------------------------ policy (dqn.py:330-375)
  def policy_draws(self, batch_size):
    """The random numbers one exploring `policy` call over `batch_size` samples consumes, drawn as that call draws them.
    A policy evaluated group by group (`PipelinedVecStackEnv.collect_step`) hands each group its slice (`draws=`) and
    takes the actions one call over the whole batch would take."""
    if self._exploration_mode != 'epsilon-greedy':
      raise NotImplementedError('group-wise collection is implemented for epsilon-greedy exploration')
    dev = next(self._q_net.parameters()).device
    u = torch.rand(batch_size, generator=self._gen, device=dev)
    return u, torch.randint(self._n_actions, (batch_size,), generator=self._gen, device=dev)

  @torch.no_grad()
  def policy(self, inputs, exploration=False, values=False, draws=None):
    timer = self.policy_timer
    if timer is not None:
      timer.start()
    try:
      return self._policy(inputs, exploration, values, draws)
    finally:
      if timer is not None:
        timer.stop()

  def _policy(self, inputs, exploration, values, draws):
    if self._policy_op is not None and exploration and not values and self._exploration_mode == 'epsilon-greedy':
      if draws is None:
        return self._policy_op(self._q_net, inputs, self.exploration, self._gen)
      return self._policy_op(self._q_net, inputs, self.exploration, self._gen, draws=draws)
    with torch.no_grad():
      q = self._q_net(inputs)
    greedy = torch.argmax(q, dim=-1)                 # ties -> lowest index
    if exploration:
      e = self.exploration
      if self._exploration_mode == 'epsilon-greedy':
        u, rnd = draws if draws is not None else self.policy_draws(q.shape[0])
        actions = torch.where(u > e, greedy, rnd)
      else:
        z = -torch.log(-torch.log(torch.rand(q.shape, generator=self._gen, device=q.device)))
        actions = torch.argmax(q / e + z, dim=-1)
    else:
      actions = greedy
    return (actions, q) if values else actions

  def acknowledge_reset(self):
    self._replay_memory.set_terminal()               # dqn.py:381-385

  def observe(self, state, reward, terminal, action):
    self._replay_memory.add(state, reward, terminal, action)

  def collect(self, state, reward, terminal):
    action = self.policy(state, exploration=True)
    self._replay_memory.add(state, reward, terminal, action)
    return action

  # ------------------------------------------------------------------ train (dqn.py:397-486)
  def td_targets(self, rewards, next_states, terminal):
    """y = r + where(terminal, 0, gamma * Q_target(s', argmax_a Q(s', a)))  (dqn.py:418-454)."""
    with torch.no_grad():
      if self._reward_scale is not None:
        rewards = rewards * self._reward_scale
      if self._gamma == 0 and not self._n_step:
        return rewards
      tq = self._g_target(next_states) if self._graphs else self._target_q_net(next_states)
      if self._double:
        a = torch.argmax(self._g_online(next_states) if self._graphs else self._q_net(next_states), dim=-1)
        tq = tq.gather(1, a[:, None])[:, 0]
      else:
        tq = tq.amax(dim=-1)
      if self._n_step:
        rewards = (self._gamma_r * rewards).sum(dim=-1)
      return rewards + torch.where(terminal, torch.zeros_like(tq), self._gamma * tq)

  def loss_from_td(self, td_abs, weights=None):
    if self._huber:
      quadratic = torch.clamp(td_abs, max=self._huber_delta)
      linear = td_abs - quadratic
      loss = 0.5 * quadratic ** 2 + self._huber_delta * linear       # dqn.py:461-464
    else:
      loss = 0.5 * td_abs ** 2
    if weights is not None:
      loss = loss * weights
    return loss.mean()

  def _draw(self):
    """One minibatch from the replay memory as a flat tuple of tensors:
    (indexes | None, weights | None, states..., actions, rewards, next_states..., terminal)."""
    if self._prioritized:
      indexes, weights, (states, actions, rewards, next_states, terminal) = \
        self._replay_memory.sample(self._minibatch_size, get_weights=True)
    else:
      indexes = weights = None
      states, actions, rewards, next_states, terminal = self._replay_memory.sample(self._minibatch_size)
    return (indexes, weights) + tuple(states) + (actions, rewards) + tuple(next_states) + (terminal,)

  def _next_minibatch(self):
    """`next(self._replay_memory_iter)` (dqn.py:399-405).  The reference reads its minibatches through
    `dataset.prefetch(prefetch)` (dqn.py:247-252; config.gin:104 sets 3): a background thread keeps `prefetch` minibatches
    sampled AHEAD of the update that consumes them, so a minibatch was drawn — with the priorities, the importance-weight
    schedule and the transitions of that moment — about `prefetch` updates before it is used.  Restated deterministically:
    a FIFO of `prefetch` minibatches, filled on the first call; every update takes the oldest and a new one is drawn in its
    place before the update's own priorities are written (the producer thread refills as soon as a slot is free).  The slots
    live at fixed addresses (shift copies), so the whole thing replays from the update's hipGraph."""
    k = self._prefetch
    if not k:
      flat = self._draw()
    else:
      flat = self._peek_minibatch()
      self._advance_fifo()
    return self._split_minibatch(flat)

  def _peek_minibatch(self):
    """A copy of the minibatch at the head of the FIFO (filled with `prefetch` draws on the first call)."""
    if self._fifo is None:
      self._fifo = [tuple(None if t is None else t.clone() for t in self._draw()) for _ in range(self._prefetch)]
    return tuple(None if t is None else t.clone() for t in self._fifo[0])

  def _advance_fifo(self):
    """The head leaves, a new minibatch is drawn into the tail (slots at fixed addresses: shift copies)."""
    k = self._prefetch
    for i in range(k - 1):
      for dst, src in zip(self._fifo[i], self._fifo[i + 1]):
        if dst is not None:
          dst.copy_(src)
    for dst, src in zip(self._fifo[k - 1], self._draw()):
      if dst is not None:
        dst.copy_(src)

  @staticmethod
  def _split_minibatch(flat):
    indexes, weights = flat[0], flat[1]
    ns = (len(flat) - 5) // 2
    states, actions, rewards = flat[2:2 + ns], flat[2 + ns], flat[3 + ns]
    next_states, terminal = flat[4 + ns:4 + 2 * ns], flat[4 + 2 * ns]
    return indexes, weights, (states, actions, rewards, next_states, terminal)

  def _forward_backward(self, peek=False):
    """First half of one minibatch update (dqn.py:397-469): sample, target evaluations, forward, loss, backward into the
    flat gradient bucket.  Returns (loss, mean TD error, indexes, |TD|, new priorities or None).
    peek: take the head of the FIFO and leave the FIFO as it is (`train_begin`; `_advance_fifo` follows in `train`)."""
    indexes, weights, (states, actions, rewards, next_states, terminal) = \
      self._split_minibatch(self._peek_minibatch()) if peek else self._next_minibatch()
    if not self._bias_compensation:
      weights = None
    self._last_sample_indexes = indexes
    new_logits = None
    if self._fused and not (self._gamma == 0 and not self._n_step):
      # target evaluations, then loss, mean TD, |TD|, new priorities and d loss / d Q(s, .) in one kernel (dqn.py:408-469)
      from stackrl_amd import qops
      if self._hand is not None:
        # hand-written convolutions: Q_target(s', .), then Q(s, .) and Q(s', .) of the online net in ONE pass over the 2 mb
        # samples (activations saved; the backward runs over the first mb)
        mb = int(actions.shape[0])
        self._hand.refresh()
        with torch.no_grad():
          tq = self._hand_t.forward(next_states)
          if self._n_step:
            rewards = (self._gamma_r * rewards).sum(dim=-1)
        if self._double:
          both = tuple(torch.cat([a, b]) for a, b in zip(states, next_states))
          q2 = self._hand.forward(both, save=True)
          q_all, qo = q2[:mb], q2[mb:].detach()
        else:
          q_all, qo = self._hand.forward(states, save=True), None
        loss, mtd, td_abs, new_logits, grad_q = qops.td_epilogue(
          q_all.detach(), qo, tq, actions, rewards, terminal, weights, self._gamma, self._huber_delta, self._reward_scale,
          self._double, self._replay_memory.epsilon, self._ws)
        # no zero-fill of the gradient bucket: the hand-written backward writes every element of every parameter's gradient
        # (tests/test_train_conv_gpu.py starts it from NaN); the alignment padding between parameters keeps its initial zeros
        first = not getattr(self, '_hand_cover_checked', False) and not torch.cuda.is_current_stream_capturing()
        if first:
          # once, on the first eager update: every parameter's gradient starts from NaN, so that a parameter the hand-written
          # backward does not write (a layer HandNet does not cover, a frozen or non-dueling variant) cannot keep the
          # previous step's — or, with several ranks, the previously all-reduced — value unnoticed
          for p in self._q_net.parameters():
            p.grad.fill_(float('nan'))
        self._hand.backward(grad_q)
        if first:
          self._hand_cover_checked = True
          missed = [n for n, p in self._q_net.named_parameters() if not bool(torch.isfinite(p.grad).all())]
          if missed:
            raise RuntimeError('HandNet.backward left (part of) these gradients unwritten: {}'.format(missed))
        return loss, mtd, indexes, td_abs, new_logits
      with torch.no_grad():
        tq = self._g_target(next_states) if self._graphs else self._target_q_net(next_states)
        qo = (self._g_online(next_states) if self._graphs else self._q_net(next_states)) if self._double else None
        if self._n_step:
          rewards = (self._gamma_r * rewards).sum(dim=-1)
      q_all = self._q_net(states)
      loss, mtd, td_abs, new_logits, grad_q = qops.td_epilogue(
        q_all.detach(), qo, tq, actions, rewards, terminal, weights, self._gamma, self._huber_delta, self._reward_scale,
        self._double, self._replay_memory.epsilon, self._ws)
      self._flat_grad.zero_()
      q_all.backward(grad_q)
    else:
      y = self.td_targets(rewards, next_states, terminal)
      q = self._q_net(states).gather(1, actions[:, None])[:, 0]      # one_hot . sum, dqn.py:410-417
      td = q - y
      mtd = td.mean().detach()
      td_abs = td.abs()
      loss = self.loss_from_td(td_abs, weights)
      self._flat_grad.zero_()
      loss.backward()
      loss, td_abs = loss.detach(), td_abs.detach()
    return loss, mtd, indexes, td_abs, new_logits

  def _all_reduce(self):
    """The one collective of the update: the flat gradient bucket summed over the ranks (RCCL over xGMI)."""
    dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM, group=self._pg)

  def _apply(self, indexes, td_abs, new_logits):
    """Second half (dqn.py:470-476): average over the ranks, optimiser step, new replay priorities."""
    if self._world > 1:
      self._flat_grad.div_(self._world)
    if isinstance(self._optimizer, KerasAdam):
      self._optimizer.step(self._flat_grad)
    else:
      self._optimizer.step()
    if self._prioritized:
      self._replay_memory.update_priorities(indexes, td_abs, logits=new_logits)   # dqn.py:475-476

  def _update(self):
    """One minibatch update (dqn.py:397-476) without the host-side bookkeeping; returns (loss, mean TD error)."""
    loss, mtd, indexes, td_abs, new_logits = self._forward_backward()
    if self._world > 1:
      self._all_reduce()
    self._apply(indexes, td_abs, new_logits)
    return loss, mtd

  _GRAPH_WARMUP = 3   # eager updates before the capture (library solver search, optimiser state, lazy initialisations)

  def train_begin(self):
    """Start the gradient half of the next update beside the collect step (see `early_gradient`).  Call it before
    `collect` / `Trainer.collect_step`; `train()` completes the update.  Returns whether anything was started."""
    if not self._early or self._pending is not None or self._fifo is None:
      return False                           # (the first update fills the FIFO from the replay memory: serial)
    if self._graphs and self._grad_graph is None and self._graph_calls < self._GRAPH_WARMUP:
      return False                           # eager warm-up updates run serially
    if self._hand_t is not None:
      self._hand_t.refresh_if_stale()
    main, upd = torch.cuda.current_stream(self.device), self._upd_stream
    self._replay_memory.refresh_schedules()
    upd.wait_stream(main)                    # the weights of the previous update, the FIFO as the previous train() left it
    with torch.cuda.stream(upd):
      if not self._graphs:
        out = self._forward_backward(peek=True)
        for t in out:
          if torch.is_tensor(t):
            t.record_stream(main)            # allocated on the update stream, consumed by train() on the current one
      else:
        if self._grad_graph is None:
          self._capture_early()
        self._grad_graph.replay()
        out = self._graph_out_early
    self._pending = out
    return True

  def _capture_early(self):
    """The early form of the graph-replayed update: one graph for the gradient half (captured and replayed on the update
    stream), one for the rest (FIFO shift + draw, optimiser step, priorities) on the current stream; with more than one
    rank the all-reduce sits between them as an ordinary stream-ordered call."""
    mem = self._replay_memory
    mem.tensor_schedules = True
    g_eval, self._graphs = self._graphs, False       # the target evaluations are part of the graph
    try:
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g, stream=self._upd_stream, capture_error_mode=CAPTURE_MODE):
        self._graph_out_early = self._forward_backward(peek=True)
      self._grad_graph = g
    finally:
      self._graphs = g_eval

  def _finish_early(self):
    loss, mtd, indexes, td_abs, new_logits = self._pending
    self._pending = None
    torch.cuda.current_stream(self.device).wait_stream(self._upd_stream)
    if not self._graphs:
      self._advance_fifo()
      if self._world > 1:
        self._all_reduce()
      self._apply(indexes, td_abs, new_logits)
      return loss, mtd
    mem = self._replay_memory
    if self._tail_graph is None:
      g = torch.cuda.CUDAGraph()
      g.register_generator_state(mem._gen)
      with torch.cuda.graph(g, pool=self._grad_graph.pool(), capture_error_mode=CAPTURE_MODE):
        self._advance_fifo()
      self._fifo_graph = g
      g2 = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g2, pool=self._grad_graph.pool(), capture_error_mode=CAPTURE_MODE):
        self._apply(indexes, td_abs, new_logits)
      self._tail_graph = g2
    self._fifo_graph.replay()
    if self._world > 1:
      self._all_reduce()
    self._tail_graph.replay()
    return loss.clone(), mtd.clone()

  def train(self):
    if self._pending is not None:
      loss, mtd = self._finish_early()
    else:
      if self._hand_t is not None:
        self._hand_t.refresh_if_stale()        # eagerly, outside any graph: the target net changes only through framework ops
      if self._graphs and not (self._early and self._fifo is not None and self._graph_calls >= self._GRAPH_WARMUP):
        loss, mtd = self._train_graphed()
      else:
        if self._graphs:
          self._replay_memory.refresh_schedules()
          self._graph_calls += 1
        loss, mtd = self._update()
    self._iterations += 1
    # consumers that cache re-packed weights (qops.FastFeatures) key on this: a graph replay changes the parameters
    # without bumping any tensor version
    self._q_net._weights_epoch = getattr(self._q_net, '_weights_epoch', 0) + 1   # monotonic: bumped by every update and every restore
    if self._iterations % self._target_update_period == 0:           # dqn.py:478-484
      self._target_q_net.load_state_dict(self._target_sync_source())
      if self._hand_t is not None:
        self._hand_t.refresh()                                       # eagerly: the target's packed weights are read by the graph
    return loss, mtd

  def _target_sync_source(self):
    return self._q_net.state_dict()

  def _train_graphed(self):
    """The whole update — prioritised sampling, target evaluation, forward, backward, Adam, priority update: hundreds of
    launches of mostly microsecond kernels — replayed as one hipGraph.  Everything it reads or writes lives at fixed
    addresses (replay tensors, trackers, flat gradient and parameter buckets, Adam state, schedule scalars), the sampling
    generator is registered with the graph, and the nets' parameters are updated in place, so target syncs and
    checkpoints need no re-capture.  With more than one rank the update is two graphs around the eager all-reduce of the
    gradient bucket (sample .. backward | all-reduce | average, Adam, priorities): the collective stays an ordinary
    stream-ordered call of whatever backend the process group has, and the launch-bound halves are still one launch each."""
    mem = self._replay_memory
    mem.refresh_schedules()
    if self._train_graph is None:
      self._graph_calls += 1
      if self._graph_calls <= self._GRAPH_WARMUP:
        return self._update()
      mem.tensor_schedules = True
      g = torch.cuda.CUDAGraph()
      g.register_generator_state(mem._gen)
      g_eval = self._graphs
      self._graphs = False                 # the target evaluations are part of this graph, not graphs of their own
      try:
        if self._world == 1:
          with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            self._graph_out = self._update()
        else:
          with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            loss, mtd, indexes, td_abs, new_logits = self._forward_backward()
          self._graph_out = (loss, mtd)
          g2 = torch.cuda.CUDAGraph()
          with torch.cuda.graph(g2, pool=g.pool(), capture_error_mode=CAPTURE_MODE):
            self._apply(indexes, td_abs, new_logits)
          self._apply_graph = g2
      finally:
        self._graphs = g_eval
      self._train_graph = g
    return self._replay_update()

  def _replay_update(self):
    self._train_graph.replay()
    if self._world > 1:
      self._all_reduce()
      self._apply_graph.replay()
    loss, mtd = self._graph_out
    return loss.clone(), mtd.clone()

  def save_weights(self, path):
    torch.save(self._q_net.state_dict(), path)

  def state_dict(self, memory=True):
    """Everything `tf.train.Checkpoint(agent=...)` tracks in the reference (training.py:199-208): both nets, the
    optimiser slots, the iteration counter, the RNG stream and (optionally) the replay memory."""
    d = {'q_net': self._q_net.state_dict(), 'target_q_net': self._target_q_net.state_dict(),
         'optimizer': self._optimizer.state_dict(), 'iterations': int(self._iterations), 'gen': self._gen.get_state()}
    if memory:
      d['replay_memory'] = self._replay_memory.state_dict()
      # the minibatches already drawn and waiting (`dataset.prefetch`, dqn.py:247-252): part of the state, or a resumed run
      # would train on freshly drawn ones where the uninterrupted run trains on these
      if self._fifo is not None:
        d['prefetched'] = [tuple(None if t is None else t.clone() for t in slot) for slot in self._fifo]
    return d

  def load_state_dict(self, d):
    """Accepts the whole dict of `state_dict` or a part of it (the Trainer keeps what the replicas share and what belongs
    to one rank in separate files)."""
    if 'q_net' in d:
      self._q_net.load_state_dict(d['q_net'])
      self._target_q_net.load_state_dict(d['target_q_net'])
      self._optimizer.load_state_dict(d['optimizer'])
      self._iterations = int(d['iterations'])
      self._q_net._weights_epoch = getattr(self._q_net, '_weights_epoch', 0) + 1   # restored weights: a new epoch, never an earlier one's
      if self._hand_t is not None:
        self._hand_t.refresh()
    if self._pending is not None:
      # a gradient half computed from the weights / minibatch being replaced is in flight on the update stream: wait for
      # it and drop it (the next train() starts over from the restored state)
      torch.cuda.current_stream(self.device).wait_stream(self._upd_stream)
      self._pending = None
    if 'gen' in d:
      self._gen.set_state(d['gen'].cpu())
    if 'replay_memory' in d:
      self._replay_memory.load_state_dict(d['replay_memory'])
      saved = d.get('prefetched')
      graphed = self._train_graph is not None or self._grad_graph is not None
      if saved is not None and len(saved) == self._prefetch:
        # the waiting minibatches of the saved run, into the slots a captured update reads at their addresses (or new ones)
        if self._fifo is None:
          self._fifo = [tuple(None if t is None else t.clone().to(self.device) for t in slot) for slot in saved]
        else:
          for slot, src in zip(self._fifo, saved):
            for dst, t in zip(slot, src):
              if dst is not None:
                dst.copy_(t)
      elif self._fifo is not None and not graphed:
        self._fifo = None                      # minibatches drawn from the memory that was just replaced
      elif self._fifo is not None:             # (a captured update reads the slots at their addresses: refill in place)
        for slot in self._fifo:
          for dst, src in zip(slot, self._draw()):
            if dst is not None:
              dst.copy_(src)

  def reseed(self, seed):
    """New streams for exploration and minibatch sampling (a rank that resumes without state of its own)."""
    self._gen.manual_seed(int(seed) % (2 ** 63) + 1)
    self._replay_memory._gen.manual_seed(int(seed) % (2 ** 63))