"""GPU tests of the learner rows: the hand-written Q-net ops against a plain PyTorch fp32 reference, the fused
rollout policy against `DQN.policy`, and an end-to-end collect -> step -> train loop on the HIP env."""
import numpy as np
import pytest

import math
torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,C,H,h', [(3, 16, 128, 32), (2, 16, 64, 16), (1, 5, 40, 32)])
def test_xcorr_hip_matches_torch_fp32(B, C, H, h):
  from stackrl_amd import nets, qops
  g = torch.Generator(device='cuda').manual_seed(B * 7 + C)
  x = torch.rand((B, C, H, H), generator=g, device='cuda')          # post-ReLU features are non-negative
  w = torch.rand((B, C, h, h), generator=g, device='cuda') - 0.3
  ref = nets.correlation_reference(x.double(), w.double()).float()  # fp64 accumulate reference
  for precision in (None, 'fp32'):                                  # default route (MFMA bf16x3 where built) and the vector kernel
    got = qops.xcorr_forward(x, w, precision)
    assert got.shape == ref.shape == (B, 1, H - h + 1, H - h + 1)
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
  # fp32 accumulation of C*h*h <= 16,384 products: stated tolerance 2e-5 relative to the result scale
  scale = float(ref.abs().max())
  assert float((got - ref).abs().max()) <= 2e-5 * scale
  ref32 = nets.correlation_reference(x, w)
  assert float((got - ref32).abs().max()) <= 5e-5 * scale


@pytest.mark.parametrize('B,C,H,h', [(3, 16, 128, 32), (5, 16, 64, 16), (2, 7, 128, 32)])
def test_xcorr_mfma_bf16_matches_torch(B, C, H, h):
  """MFMA path of the cross-correlation (csrc/xcorr_mfma.hip): bf16 products are exact in fp32, so against an fp64
  accumulation of the same bf16-rounded operands only the fp32 accumulation order differs."""
  from stackrl_amd import nets, qops
  g = torch.Generator(device='cuda').manual_seed(B * 11 + C)
  x = torch.rand((B, C, H, H), generator=g, device='cuda').to(torch.bfloat16)
  w = (torch.rand((B, C, h, h), generator=g, device='cuda') - 0.3).to(torch.bfloat16)
  got = qops.xcorr_forward(x, w)
  assert got.dtype == torch.float32 and got.shape == (B, 1, H - h + 1, H - h + 1)
  ref = nets.correlation_reference(x.double(), w.double()).float()
  scale = float(ref.abs().max())
  assert float((got - ref).abs().max()) <= 2e-5 * scale          # stated tolerance: fp32 accumulation of <= 16,384 terms
  # against the fp32 vector kernel on the un-rounded operands the difference is the bf16 input rounding (2^-9 relative
  # per operand, averaging out over the sum)
  x32 = torch.rand((B, C, H, H), generator=g, device='cuda'); w32 = torch.rand((B, C, h, h), generator=g, device='cuda') - 0.3
  a = qops.xcorr_forward(x32.to(torch.bfloat16), w32.to(torch.bfloat16)); b = qops.xcorr_forward(x32, w32)
  assert float((a - b).abs().max()) <= 4e-3 * float(b.abs().max())


@pytest.mark.parametrize('B,C,H,h', [(3, 16, 128, 32), (4, 16, 64, 16), (2, 5, 128, 32)])
@pytest.mark.parametrize('precision,tol', [(0, 6e-3), (1, 2e-5)])
def test_xcorr_mfma_autograd_matches_torch_fp64(B, C, H, h, precision, tol):
  """Forward and both gradients of the MFMA cross-correlation against the library formulation in fp64.
  Stated tolerances relative to each tensor's scale: 6e-3 with operands rounded to bf16, 2e-5 (the fp32 vector
  kernel's own tolerance) with the bf16x3 split."""
  from stackrl_amd import nets, qops
  g = torch.Generator(device='cuda').manual_seed(B * 13 + C + precision)
  x = torch.rand((B, C, H, H), generator=g, device='cuda').requires_grad_()
  w = (torch.rand((B, C, h, h), generator=g, device='cuda') - 0.3).requires_grad_()
  go = torch.randn((B, 1, H - h + 1, H - h + 1), generator=g, device='cuda')
  out = qops.correlation(precision)(x, w)
  out.backward(go)
  xd = x.detach().double().requires_grad_(); wd = w.detach().double().requires_grad_()
  ref = nets.correlation_reference(xd, wd)
  ref.backward(go.double())
  for got, want in ((out, ref), (x.grad, xd.grad), (w.grad, wd.grad)):
    assert got.shape == want.shape and got.dtype == torch.float32
    err = float((got.detach().double() - want.detach()).abs().max()); scale = float(want.detach().abs().max())
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize('cin,cout,H', [(16, 16, 32), (16, 32, 16), (32, 16, 48), (32, 32, 16), (64, 32, 32), (64, 16, 16)])
def test_conv3x3_mfma_matches_torch(cin, cout, H):
  """csrc/conv_mfma.hip against F.conv2d on the same bf16-rounded operands in fp32 (bf16 products are exact in fp32;
  only the accumulation order and the final bf16 rounding differ: half a bf16 ulp = 2^-9 relative), for the plain,
  concat-slice + pooled and channel-major output modes."""
  from stackrl_amd import qops
  F = torch.nn.functional
  g = torch.Generator(device='cuda').manual_seed(cin * 3 + cout + H)
  B, W = 3, H + 16
  x = torch.rand((B, cin, H, W), generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
  w = ((torch.rand((cout, cin, 3, 3), generator=g, device='cuda') - 0.5) * 0.2)
  b = torch.rand(cout, generator=g, device='cuda') - 0.5
  wf = qops.pack_conv3x3_weights(w)
  ref = F.relu(F.conv2d(x.float(), w.to(torch.bfloat16).float(), b, padding=1))
  tol = lambda t: 2.0 ** -8 * t.abs().clamp(min=1e-2)
  y = qops.conv3x3_bias_relu(x, wf, b, cout)
  assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
  assert bool(((y.float() - ref).abs() <= tol(ref)).all())
  cat = torch.full((B, cout + 16, H, W), 7.0, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
  y2, p = qops.conv3x3_bias_relu(x, wf, b, cout, out=cat, out_offset=16, pool=True)
  assert torch.equal(cat[:, 16:], y) and bool((cat[:, :16] == 7.0).all())          # only the slice was written
  assert torch.equal(p, F.max_pool2d(y, 2))
  z = qops.conv3x3_bias_relu(x, wf, b, cout, nchw=True)
  assert z.is_contiguous() and torch.equal(z, y.contiguous())


@pytest.mark.parametrize('cin,cout,H,W', [(32, 16, 12, 32), (64, 32, 8, 16)])
def test_convt2x2_mfma_matches_torch(cin, cout, H, W):
  from stackrl_amd import qops
  F = torch.nn.functional
  g = torch.Generator(device='cuda').manual_seed(cin + H)
  B = 3
  x = torch.rand((B, cin, H, W), generator=g, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
  w = (torch.rand((cin, cout, 2, 2), generator=g, device='cuda') - 0.5) * 0.3
  b = torch.rand(cout, generator=g, device='cuda') - 0.5
  ref = F.relu(F.conv_transpose2d(x.float(), w.to(torch.bfloat16).float(), b, stride=2))
  cat = torch.full((B, 2 * cout, 2 * H, 2 * W), 3.0, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
  qops.convt2x2_bias_relu(x, qops.pack_convt2x2_weights(w), b, cout, cat, 0)
  assert bool(((cat[:, :cout].float() - ref).abs() <= 2.0 ** -8 * ref.abs().clamp(min=1e-2)).all())
  assert bool((cat[:, cout:] == 3.0).all())


def test_thin_conv_and_projection_head_match_torch():
  """The 1|2 -> 16 channel first layers (vector ALU, uint8 / 255 or float input) and the fused 16 -> 16 -> 1 tail of
  `pos_layers` (MFMA + fp32 projection) against torch in fp32."""
  from stackrl_amd import qops
  F = torch.nn.functional
  g = torch.Generator(device='cuda').manual_seed(12)
  for cin, dt in ((2, torch.uint8), (1, torch.uint8), (1, torch.float32)):
    B, H, W = 3, 40, 56
    x = torch.randint(0, 256, (B, H, W, cin), generator=g, device='cuda', dtype=torch.uint8) if dt == torch.uint8 else \
        torch.randn((B, H, W, cin), generator=g, device='cuda') * 30.0
    w = (torch.rand((16, cin, 3, 3), generator=g, device='cuda') - 0.5) * 0.5
    b = torch.rand(16, generator=g, device='cuda') - 0.5
    xf = x.float() / 255.0 if dt == torch.uint8 else x
    ref = F.relu(F.conv2d(xf.permute(0, 3, 1, 2), w, b, padding=1))
    y = qops.conv3x3_thin(x, qops.pack_thin_weights(w), b)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    assert bool(((y.float() - ref).abs() <= 2.0 ** -8 * ref.abs().clamp(min=1e-2)).all())   # fp32 math, one bf16 rounding
  # padded, zero-margined buffer: only the H x W region is written
  buf = torch.zeros((B, 16, 48, 64), dtype=torch.bfloat16, device='cuda').contiguous(memory_format=torch.channels_last)
  qops.conv3x3_thin(x, qops.pack_thin_weights(w), b, out=buf)
  assert torch.equal(buf[:, :, :H, :W], y) and float(buf[:, :, H:].abs().sum()) == 0 and float(buf[:, :, :, W:].abs().sum()) == 0
  # projection tail on the padded buffer
  w2 = (torch.rand((16, 16, 3, 3), generator=g, device='cuda') - 0.5) * 0.2
  b2 = torch.rand(16, generator=g, device='cuda') - 0.5
  pw = torch.rand(16, generator=g, device='cuda') - 0.5
  got = qops.conv3x3_relu_project(buf, qops.pack_conv3x3_weights(w2), b2, pw, 0.25, H, W)
  act = F.relu(F.conv2d(buf.float()[:, :, :H, :W], w2.to(torch.bfloat16).float(), b2, padding=1))
  ref = (act * pw[None, :, None, None]).sum(1) + 0.25
  assert got.shape == (B, H, W) and float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_fast_position_head_tracks_the_module():
  """`FastFeatures.pos` (bf16 activations, fp32 projection) against `net.pos` in fp32 on a correlation-like map: the
  stated tolerance is bf16-level (1 % of the advantage range), and the arg-max lands on a near-maximal action."""
  from stackrl_amd import nets, qops
  net = nets.DeepQSiamFCN(seed=9).cuda().eval()
  g = torch.Generator(device='cuda').manual_seed(2)
  corr = torch.randn((4, 1, 97, 97), generator=g, device='cuda') * 20.0
  got = qops.FastFeatures(net).pos(corr)
  with torch.no_grad():
    ref = net.pos(corr).flatten(1)
  assert got.shape == ref.shape == (4, 97 * 97) and got.dtype == torch.float32
  span = float(ref.max() - ref.min())
  assert float((got - ref).abs().max()) <= 1e-2 * span
  top = ref.gather(1, got.argmax(1, keepdim=True))[:, 0]
  assert bool((ref.max(1).values - top <= 2e-2 * span).all())


def test_fast_features_match_autocast_features():
  """Fused-epilogue inference forward of the U-Nets (qops.FastFeatures, csrc/epilogue.hip) against the stock module
  graph under bf16 autocast: same bf16 convolutions, the bias add / ReLU rounding points differ, so the stated
  tolerance is bf16-level: 3 % of the feature scale per element, 0.3 % on average."""
  from stackrl_amd import nets, qops
  net = nets.DeepQSiamFCN(seed=5).cuda().eval()
  g = torch.Generator(device='cuda').manual_seed(4)
  xm = torch.randint(0, 256, (6, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8)
  xo = torch.randint(0, 256, (6, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8)
  fx, fw = qops.FastFeatures(net)((xm, xo))
  with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
    rx, _, rw = net.features((xm, xo))
  with torch.no_grad():
    ex, _, ew = net.features((xm, xo))            # fp32
  for got, ref, exact in ((fx, rx, ex), (fw, rw, ew)):
    assert got.shape == ref.shape and got.dtype == torch.bfloat16 and got.is_contiguous()
    scale = float(exact.abs().max())
    d = (got.float() - ref.float()).abs()
    assert float(d.max()) <= 3e-2 * scale and float(d.mean()) <= 3e-3 * scale
    # and it is as close to the fp32 forward as the autocast graph is
    e_fast = float((got.float() - exact).abs().mean()); e_auto = float((ref.float() - exact).abs().mean())
    assert e_fast <= 1.5 * e_auto + 1e-6 * scale
  # the rollout policy built on it returns valid actions and mostly the arg-max of the fp32 net at epsilon 0
  pol = qops.FusedPolicy(autocast=torch.bfloat16)
  a = pol(net, (xm, xo), 0.0, torch.Generator(device='cuda').manual_seed(1))
  assert a.shape == (6,) and int(a.min()) >= 0 and int(a.max()) < net.n_actions


def test_policy_head_matches_torch():
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(3)
  adv = torch.randn((257, 9409), generator=g, device='cuda')
  adv[5, 100] = adv[5, 7000] = adv[5].max() + 1.0                   # tie -> lowest index (tf.argmax)
  u = torch.rand(257, generator=g, device='cuda')
  rnd = torch.randint(9409, (257,), generator=g, device='cuda')
  for eps in (0.0, 0.3, 1.0):
    got = qops.policy_head(adv, u, rnd, eps)
    ref = torch.where(u > eps, adv.argmax(-1), rnd)
    assert torch.equal(got, ref)
  assert int(qops.policy_head(adv, u, rnd, 0.0)[5]) == 100


def test_fused_policy_equals_dqn_policy(ref_pool):
  from stackrl_amd import nets, qops
  from stackrl_amd.dqn import DQN
  net = nets.DeepQSiamFCN(seed=4).cuda()
  g = torch.Generator(device='cuda').manual_seed(1)
  s = (torch.randint(0, 256, (6, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8),
       torch.randint(0, 256, (6, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8))
  a = DQN(net, exploration=0.4, collect_batch_size=6, replay_memory_size=60, seed=9)
  b = DQN(net, exploration=0.4, collect_batch_size=6, replay_memory_size=60, seed=9, policy_op=qops.FusedPolicy(chunk=4))
  for _ in range(3):
    assert torch.equal(a.policy(s, exploration=True), b.policy(s, exploration=True))


def test_end_to_end_dqn_iterations(ref_pool):
  """T1: collect -> non-blocking env.step -> train, a few iterations, on the HIP env with the reference's agent settings
  (config.gin:90-112)."""
  from stackrl_amd import env as envs, nets, qops
  from stackrl_amd.dqn import DQN, PolynomialDecay
  from stackrl_amd.training import Trainer
  B, L = 16, 4
  env = envs.make('Stack-v0', n_parallel=B, seed=3, pool=ref_pool, episode_length=L)
  net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
  agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=8, replay_memory_size=B * 16,
              discount_factor=.966667, collect_batch_size=B, exploration=PolynomialDecay(1.0, 400000, .1),
              prioritization=0.6, priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7,
              policy_op=qops.FusedPolicy())
  tr = Trainer(env, agent)
  tr.initialize(num_steps=12)
  assert len(agent._replay_memory) > 8
  w0 = [p.detach().clone() for p in net.parameters()]
  losses = tr.run(6)
  assert losses.shape == (6,) and bool(torch.isfinite(losses).all()) and agent.iterations == 6
  assert any(not torch.equal(a, b) for a, b in zip(w0, net.parameters()))
  # stored transitions are the env's own uint8 observations
  mem = agent._replay_memory
  assert mem._states[0].dtype == torch.uint8 and mem._states[0].shape[1:] == (128, 128, 2)
  assert int(mem._actions.max()) < env.n_actions
  env.close()


def test_update_with_mfma_xcorr_and_graphs_tracks_library_update(ref_pool):
  """The same seeded DQN run with (a) the library formulation of the cross-correlation, eager, and (b) the MFMA
  bf16x3 kernels with the target evaluations replayed from hipGraphs: identical replay contents and sampled
  minibatches (same RNG), losses within the fp32-class tolerance of the split, weights staying close."""
  from stackrl_amd import env as envs, nets, qops
  from stackrl_amd.dqn import DQN, PolynomialDecay
  from stackrl_amd.training import Trainer
  B, L = 8, 4
  runs = []
  for kw in (dict(), dict(xcorr='bf16x3', graphs=True)):
    env = envs.make('Stack-v0', n_parallel=B, seed=5, pool=ref_pool, episode_length=L)
    net = nets.DeepQSiamFCN(env.observation_spec, seed=2).cuda()
    agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=8, replay_memory_size=B * 16,
                discount_factor=.966667, collect_batch_size=B, exploration=1.0, prioritization=0.6,
                priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=9,
                policy_op=qops.FusedPolicy(), **kw)
    tr = Trainer(env, agent)
    tr.initialize(num_steps=10)
    losses = tr.run(5)
    runs.append((losses.clone(), [p.detach().clone() for p in net.parameters()], agent._replay_memory._actions.clone()))
    env.close()
  (la, wa, aa), (lb, wb, ab) = runs
  assert torch.equal(aa, ab)                                   # exploration 1.0: same random actions, same replay
  assert bool(torch.isfinite(lb).all())
  assert float((la - lb).abs().max()) <= 1e-3 * max(1e-6, float(la.abs().max()))
  for p, q in zip(wa, wb):
    assert float((p - q).abs().max()) <= 1e-3                  # 5 Adam steps of lr 6.25e-5: at most ~3e-4 per weight


def test_fast_rollout_follows_graph_replayed_updates(ref_pool):
  """The bf16 fast rollout (`FusedPolicy(autocast=bf16)` -> `FastFeatures`) caches re-packed weights; once the update is
  replayed from a hipGraph no tensor version counter moves any more, so the cache must follow the agent's weights epoch:
  after more than _GRAPH_WARMUP + 2 updates its features equal those of a cache built afresh from the current weights."""
  from stackrl_amd import env as envs, nets, qops
  from stackrl_amd.dqn import DQN, PolynomialDecay
  from stackrl_amd.training import Trainer
  B, L = 8, 4
  env = envs.make('Stack-v0', n_parallel=B, seed=5, pool=ref_pool, episode_length=L)
  net = nets.DeepQSiamFCN(env.observation_spec, seed=2).cuda()
  pol = qops.FusedPolicy(autocast=torch.bfloat16)
  agent = DQN(net, learning_rate=1e-3, adam_betas=(0.95, 0.95), minibatch_size=8, replay_memory_size=B * 16,
              discount_factor=.966667, collect_batch_size=B, exploration=0.5, prioritization=0.6,
              priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=9,
              policy_op=pol, xcorr='bf16x3', graphs=True)
  tr = Trainer(env, agent)
  tr.initialize(num_steps=10)
  tr.run(DQN._GRAPH_WARMUP + 4)
  assert agent._train_graph is not None                         # the last updates were graph replays
  obs = env.reset()()[0]
  w_before = [p.detach().clone() for p in net.parameters()]
  xa, wa = pol._ff(obs)
  tr.run(3)                                                     # three more replayed updates, no ATen op on the weights
  assert any(not torch.equal(a, b) for a, b in zip(w_before, net.parameters()))
  obs = env.reset()()[0]
  xb, wb = pol._ff(obs)                                         # refreshes the policy's own cache ...
  n = 0
  for m, (w, b) in pol._ff._w.items():                          # ... which must hold the current weights, bit for bit
    assert torch.equal(w, m.weight.detach().to(torch.bfloat16)) and torch.equal(b, m.bias.detach().float())
    n += 1
  assert n >= 20
  assert len(pol._ff._wf.packed()) >= 15 and len(pol._ff._wg.packed()) >= 8      # packed on use, after the update
  for m, wf in pol._ff._wf.packed():
    pack = qops.pack_conv3x3_weights if isinstance(m, torch.nn.Conv2d) else qops.pack_convt2x2_weights
    assert torch.equal(wf, pack(m.weight))
  for m, wg in pol._ff._wg.packed():
    pack = qops.pack_conv3x3_gemm_weights if isinstance(m, torch.nn.Conv2d) else qops.pack_convt2x2_weights
    assert torch.equal(wg, pack(m.weight))
  xc, wc = qops.FastFeatures(net)(obs)                          # and its features track a cache built afresh
  assert float((xb.float() - xc.float()).abs().max()) <= 0.05 * float(xc.float().abs().max())
  env.close()


def test_train_step_matches_dqn_oracle():
  """One `DQN.train` step of the product settings (full-size net, MFMA cross-correlation, PER, Double-DQN, Huber) against
  oracle/dqn_oracle.py on the same minibatch: sampled indices (Gumbel top-k over the oracle's logits and the generator's
  own uniforms), importance weights, transitions, TD targets / loss / mean TD in float64, updated priorities and the
  min / max trackers."""
  from oracle import dqn_oracle as O
  from stackrl_amd import nets
  from stackrl_amd.dqn import DQN
  B, slots, mb, gamma, alpha, beta = 4, 8, 6, 0.9, 0.6, 0.5
  net = nets.DeepQSiamFCN(seed=3).cuda()
  agent = DQN(net, learning_rate=1e-4, adam_betas=(0.95, 0.95), minibatch_size=mb, replay_memory_size=B * slots,
              discount_factor=gamma, collect_batch_size=B, exploration=0.5, prioritization=alpha,
              priority_bias_compensation=beta, double=True, seed=5, xcorr='bf16x3')
  with torch.no_grad():                                        # a target net that differs from the online one
    for p in agent._target_q_net.parameters():
      p.mul_(1.05)
  mem = agent._replay_memory
  ref = O.RefMemory(B, B * slots)
  rng = np.random.RandomState(0)
  for t in range(11):                                          # wraps the 8 slots per env; some episode ends
    sm = rng.randint(0, 256, (B, 128, 128, 2)).astype(np.uint8)
    so = rng.randint(0, 256, (B, 32, 32, 1)).astype(np.uint8)
    r = rng.normal(size=B).astype(np.float32)
    term = rng.rand(B) < 0.2
    a = rng.randint(0, net.n_actions, B)
    agent.observe((torch.from_numpy(sm).cuda(), torch.from_numpy(so).cuda()), torch.from_numpy(r).cuda(),
                  torch.from_numpy(term).cuda(), torch.from_numpy(a).cuda())
    ref.add([(sm[i], so[i]) for i in range(B)], r, term, a)
    if t in (5, 8):                                            # non-uniform priorities
      fin = np.flatnonzero(np.isfinite(np.array(ref.logits)))[:5]
      dl = rng.rand(len(fin)).astype(np.float32) * 3
      mem.update_priorities(torch.from_numpy(fin).cuda(), torch.from_numpy(dl).cuda())
      ref.update_priorities(fin.tolist(), dl.tolist())
  np.testing.assert_allclose(mem._logits.cpu().numpy(), np.array(ref.logits, np.float32), rtol=1e-6)
  assert int(mem._max_logit_index) == ref.max_idx and int(mem._min_logit_index) == ref.min_idx
  # --- the sample the update will draw
  st = mem._gen.get_state()
  u = torch.rand(mem._logits.shape, generator=mem._gen, device='cuda', dtype=torch.float32).cpu().numpy().astype(np.float64)
  mem._gen.set_state(st)
  lg = np.array(ref.logits, np.float64)
  keys = np.where(np.isinf(lg), lg, alpha * lg) - np.log(-np.log(u))          # memory.py:220-223
  expect_idx = np.argsort(-keys)[:mb]
  indexes, weights, (states, actions, rewards, next_states, terminal) = mem.sample(mb, get_weights=True)
  mem._gen.set_state(st)
  idx = indexes.cpu().numpy()
  assert np.array_equal(idx, expect_idx)
  for j, i in enumerate(idx.tolist()):
    s0, a0, r0, s1, t1 = ref.transition(i)
    assert np.array_equal(states[0][j].cpu().numpy(), s0[0]) and np.array_equal(states[1][j].cpu().numpy(), s0[1])
    assert np.array_equal(next_states[0][j].cpu().numpy(), s1[0]) and np.array_equal(next_states[1][j].cpu().numpy(), s1[1])
    assert int(actions[j]) == a0 and float(rewards[j]) == np.float32(r0) and bool(terminal[j]) == t1
    assert abs(float(weights[j]) - ref.weight(i, alpha, beta)) <= 1e-5 * ref.weight(i, alpha, beta)
  with torch.no_grad():
    q = agent._q_net(states).double().cpu().numpy()
    qo = agent._q_net(next_states).double().cpu().numpy()
    qt = agent._target_q_net(next_states).double().cpu().numpy()
  w = [ref.weight(i, alpha, beta) for i in idx.tolist()]
  eloss, emtd, etd = O.dqn_targets(q, qo, qt, actions.cpu().numpy(), rewards.cpu().numpy(), terminal.cpu().numpy(), gamma,
                                   double=True, huber_delta=1.0, weights=w)
  w0 = [p.detach().clone() for p in net.parameters()]
  loss, mtd = agent.train()                                    # samples the same minibatch (generator state restored)
  scale = max(1.0, float(np.abs(q).max()))
  assert abs(float(loss) - eloss) <= 2e-4 * max(eloss, 1e-3) + 1e-6 * scale
  assert abs(float(mtd) - emtd) <= 2e-4 * scale
  assert any(not torch.equal(a, b) for a, b in zip(w0, net.parameters()))
  # priorities and trackers after the update (memory.py:266-316)
  ref.update_priorities(idx.tolist(), [float(np.float32(t)) for t in etd])
  np.testing.assert_allclose(mem._logits[indexes].cpu().numpy(), np.array([ref.logits[i] for i in idx]), rtol=0, atol=2e-3)
  assert int(mem._max_logit_index) == ref.max_idx and int(mem._min_logit_index) == ref.min_idx


# ---------------------------------------------------------------------------- update-path kernels (csrc/learner.hip)
@pytest.mark.parametrize('double,huber,use_w', [(True, 1.0, True), (False, 1.0, False), (True, None, True), (True, 0.05, False)])
def test_td_epilogue_matches_torch_autograd(double, huber, use_w):
  """Loss, mean TD, |TD|, new priorities and d loss / d Q of dqn.py:408-476: the fused kernel against the same
  expressions in torch fp32 with autograd (and the loss against float64)."""
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(3)
  mb, A, gamma, eps = 32, 9409, 0.966667, 1e-3
  q = (torch.randn(mb, A, generator=g, device='cuda') * 2).requires_grad_(True)
  qo = torch.randn(mb, A, generator=g, device='cuda')
  qt = torch.randn(mb, A, generator=g, device='cuda')
  qo[3, 17] = qo[3, 4000] = 9.0                                # a tie: the lowest index wins (like argmax)
  a = torch.randint(A, (mb,), generator=g, device='cuda')
  r = torch.randn(mb, generator=g, device='cuda')
  term = torch.rand(mb, generator=g, device='cuda') < 0.3
  w = torch.rand(mb, generator=g, device='cuda') if use_w else None
  loss, mtd, td_abs, logits, grad_q = qops.td_epilogue(q.detach(), qo if double else None, qt, a, r, term, w, gamma, huber, None,
                                                       double, eps, {})
  astar = (qo if double else qt).argmax(-1)
  assert int(astar[3]) == 17 or not double
  y = r + torch.where(term, torch.zeros_like(r), gamma * qt.gather(1, astar[:, None])[:, 0])
  td = q.gather(1, a[:, None])[:, 0] - y
  ad = td.abs()
  if huber is not None:
    quad = torch.clamp(ad, max=huber); l = 0.5 * quad ** 2 + huber * (ad - quad)
  else:
    l = 0.5 * ad ** 2
  if w is not None:
    l = l * w
  ref = l.mean()
  ref.backward()
  assert abs(float(loss) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
  assert abs(float(mtd) - float(td.mean())) <= 1e-6
  assert torch.equal(td_abs, ad.detach())
  np.testing.assert_allclose(logits.cpu().numpy(), torch.log(ad.detach() + eps).cpu().numpy(), rtol=2e-6, atol=2e-6)
  np.testing.assert_allclose(grad_q.cpu().numpy(), q.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)
  assert int((grad_q != 0).sum()) <= mb


def test_adam_step_is_keras_adam():
  """`srl_adam_step` against Keras' Adam formula in float64 over several steps (epsilon outside the bias correction), and
  against the CPU path of `KerasAdam` (the same formula in torch)."""
  from stackrl_amd.dqn import KerasAdam
  rng = np.random.RandomState(0)
  n, lr, b1, b2, eps = 100003, 6.25e-5, 0.95, 0.95, 1e-7        # an odd size: exercises the scalar tail
  p0 = rng.normal(size=n).astype(np.float32)
  pg = torch.nn.Parameter(torch.from_numpy(p0.copy()).cuda())
  pc = torch.nn.Parameter(torch.from_numpy(p0.copy()))
  og, oc = KerasAdam([pg], lr, (b1, b2), eps), KerasAdam([pc], lr, (b1, b2), eps)
  p = p0.astype(np.float64); m = np.zeros(n); v = np.zeros(n)
  for t in range(1, 8):
    gnp = (rng.normal(size=n) * (10.0 ** rng.uniform(-4, 1))).astype(np.float32)
    gfull = torch.zeros(og.flat.numel()); gfull[:n] = torch.from_numpy(gnp)   # the bucket pads every parameter to 256 bytes
    og.step(gfull.cuda()); oc.step(gfull)
    g = gnp.astype(np.float64)
    m += (g - m) * (1 - b1); v += (g * g - v) * (1 - b2)
    p -= lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps)
  np.testing.assert_allclose(pg.detach().cpu().numpy(), p, rtol=0, atol=2e-6)
  np.testing.assert_allclose(pc.detach().numpy(), p, rtol=0, atol=2e-6)
  assert float(og.state[0]) == 7.0 and pg.data_ptr() == og.flat.data_ptr()
  # the kernel itself on an odd length (scalar tail), one step from zero state: p -= lr sqrt(1-b2)/(1-b1) m / (sqrt(v) + eps)
  from stackrl_amd import qops
  n2 = 1027
  P = torch.from_numpy(p0[:n2].copy()).cuda(); G = torch.from_numpy(gnp[:n2].copy()).cuda()
  M = torch.zeros(n2, device='cuda'); V = torch.zeros(n2, device='cuda'); S = torch.tensor([0., 1., 1., 0.], device='cuda')
  qops.adam_step(P, G, M, V, S, lr, b1, b2, eps)
  g = gnp[:n2].astype(np.float64)
  ref = p0[:n2].astype(np.float64) - lr * np.sqrt(1 - b2) / (1 - b1) * (g * (1 - b1)) / (np.sqrt(g * g * (1 - b2)) + eps)
  np.testing.assert_allclose(P.cpu().numpy(), ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize('n,k', [(1000, 8), (65536, 32), (262144, 32), (5000, 64)])
def test_gumbel_topk_matches_torch_topk(n, k):
  """K7 against `torch.topk` of alpha * logit + Gumbel(u) (memory.py:220-223): same indices in the same order wherever
  the float32 keys are not within rounding of each other, unsampleable (-inf) slots never returned."""
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(n + k)
  logits = torch.randn(n, generator=g, device='cuda') * 2
  logits[torch.rand(n, generator=g, device='cuda') < 0.3] = -math.inf
  u = torch.rand(n, generator=g, device='cuda')
  alpha = torch.tensor(0.6, device='cuda')
  idx, key = qops.gumbel_topk(logits, u, alpha, k, {})
  keys = torch.where(torch.isinf(logits), logits, alpha * logits) - torch.log(-torch.log(u))
  tv, ti = torch.topk(keys, k)
  assert bool(torch.isfinite(logits[idx]).all()) and len(set(idx.tolist())) == k
  np.testing.assert_allclose(key.cpu().numpy(), tv.cpu().numpy(), rtol=1e-5, atol=1e-5)
  np.testing.assert_allclose(keys[idx].cpu().numpy(), key.cpu().numpy(), rtol=1e-5, atol=1e-5)
  gaps = (tv[:-1] - tv[1:]).min()
  if float(gaps) > 1e-4:
    assert torch.equal(idx, ti)
  # fewer sampleable slots than k: the missing entries are flagged by a -inf key
  logits2 = torch.full((n,), -math.inf, device='cuda'); logits2[5] = 0.0; logits2[n - 1] = 1.0
  idx2, key2 = qops.gumbel_topk(logits2, u, alpha, k, {})
  assert sorted(idx2[:2].tolist()) == [5, n - 1] and bool(torch.isinf(key2[2:]).all()) and bool(torch.isfinite(key2[:2]).all())


@pytest.mark.parametrize('literal', [False, True])
def test_replay_kernels_match_the_library_formulation(literal):
  """K8 (scatter of `add`, gather of `sample`) + K7 inside `ReplayMemory`: a memory on the hand-written kernels against one
  forced onto the index_copy / topk / gather formulation, same data and same generator seed."""
  from stackrl_amd.memory import ReplayMemory
  B, slots, mb = 6, 9, 8
  spec = (((B, 128, 128, 2), torch.uint8), ((B, 32, 32, 1), torch.uint8))
  mk = lambda: ReplayMemory(spec, B * slots, alpha=0.6, beta=0.7, seed=11, device='cuda', reference_next_index=literal)
  a, b = mk(), mk()
  assert a._fused
  b._fused = False
  g = torch.Generator(device='cuda').manual_seed(1)
  for t in range(13):
    st = (torch.randint(0, 256, (B, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8),
          torch.randint(0, 256, (B, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8))
    r = torch.randn(B, generator=g, device='cuda')
    term = torch.rand(B, generator=g, device='cuda') < 0.2
    act = torch.randint(0, 9409, (B,), generator=g, device='cuda')
    a.add(st, r, term, act); b.add(st, r, term, act)
    if t in (6, 9):
      ia, wa, _ = a.sample(4, get_weights=True); ib, wb, _ = b.sample(4, get_weights=True)
      d = torch.rand(4, generator=g, device='cuda') * 2
      a.update_priorities(ia, d); b.update_priorities(ib, d)
  for k in ('_rewards', '_terminal', '_actions', '_logits', '_max_logit', '_min_logit', '_max_logit_index', '_min_logit_index'):
    assert torch.equal(getattr(a, k), getattr(b, k)), k
  assert all(torch.equal(x, y) for x, y in zip(a._states, b._states))
  ia, wa, (sa, aa, ra, na, ta) = a.sample(mb, get_weights=True)
  ib, wb, (sb, ab, rb, nb_, tb) = b.sample(mb, get_weights=True)
  assert torch.equal(ia, ib) and torch.equal(aa, ab) and torch.equal(ra, rb) and torch.equal(ta, tb)
  assert all(torch.equal(x, y) for x, y in zip(sa + na, sb + nb_))
  np.testing.assert_allclose(wa.cpu().numpy(), wb.cpu().numpy(), rtol=2e-6)


@pytest.mark.parametrize('cin,cout,H', [(16, 16, 32), (16, 32, 16), (32, 16, 48), (32, 32, 16), (64, 32, 32), (64, 16, 16)])
def test_conv3x3_bf16x3_matches_torch_fp64(cin, cout, H):
  """The fp32-class MFMA convolution (`k_conv3x3_x3`: bf16 hi/lo split, hi hi + hi lo + lo hi, fp32 accumulation) + bias +
  ReLU against float64 torch: 3e-5 of the output scale (the split's 2^-16 per product; fp32 itself sits at 1e-6 here), for
  the plain, concat-slice + pooled and channel-major outputs."""
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(cin * 100 + cout)
  B, W = 3, H
  x = torch.randn(B, cin, H, W, generator=g, device='cuda').contiguous(memory_format=torch.channels_last)
  w = torch.randn(cout, cin, 3, 3, generator=g, device='cuda') / (3 * cin ** 0.5)
  b = torch.randn(cout, generator=g, device='cuda') * 0.1
  ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
  wf = qops.pack_conv3x3_weights_x3(w)
  y = qops.conv3x3_bias_relu(x, wf, b, cout)
  scale = float(ref.abs().max())
  assert y.dtype == torch.float32 and float((y.double() - ref).abs().max()) <= 3e-5 * scale
  cat = torch.zeros((B, 2 * cout, H, W), device='cuda').contiguous(memory_format=torch.channels_last)
  _, pooled = qops.conv3x3_bias_relu(x, wf, b, cout, out=cat, out_offset=cout, pool=True)
  assert float((cat[:, cout:].double() - ref).abs().max()) <= 3e-5 * scale and float(cat[:, :cout].abs().max()) == 0.0
  assert torch.equal(pooled, torch.nn.functional.max_pool2d(cat[:, cout:], 2))
  yn = qops.conv3x3_bias_relu(x, wf, b, cout, nchw=True)
  assert yn.is_contiguous() and torch.equal(yn, y.contiguous())


@pytest.mark.parametrize('cin,cout,H,W', [(32, 16, 12, 32), (64, 32, 8, 16)])
def test_convt2x2_bf16x3_matches_torch_fp64(cin, cout, H, W):
  """The fp32-class transposed convolution (`k_convt2x2_x3`) + bias + ReLU into a slice of the concat buffer against
  float64 torch: 3e-5 of the output scale; the other half of the buffer untouched."""
  from stackrl_amd import qops
  F = torch.nn.functional
  g = torch.Generator(device='cuda').manual_seed(cin + H)
  B = 3
  x = torch.randn((B, cin, H, W), generator=g, device='cuda').contiguous(memory_format=torch.channels_last)
  w = torch.randn((cin, cout, 2, 2), generator=g, device='cuda') / cin ** 0.5
  b = torch.randn(cout, generator=g, device='cuda') * 0.1
  ref = F.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2))
  cat = torch.full((B, 2 * cout, 2 * H, 2 * W), 3.0, device='cuda').contiguous(memory_format=torch.channels_last)
  qops.convt2x2_bias_relu(x, qops.pack_convt2x2_weights_x3(w), b, cout, cat, 0)
  assert float((cat[:, :cout].double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
  assert bool((cat[:, cout:] == 3.0).all())


def test_thin_conv_and_projection_head_fp32():
  """float32 outputs of the thin first layers (fp32 FMAs: 1e-6 of the scale against float64) and the fp32-class
  16 -> 16 -> 1 tail of `pos_layers` (bf16x3 products, fp32 projection: 3e-5) on a zero-margined padded buffer."""
  from stackrl_amd import qops
  F = torch.nn.functional
  g = torch.Generator(device='cuda').manual_seed(21)
  B, H, W = 3, 40, 56
  for cin, dt in ((2, torch.uint8), (1, torch.uint8), (1, torch.float32)):
    x = torch.randint(0, 256, (B, H, W, cin), generator=g, device='cuda', dtype=torch.uint8) if dt == torch.uint8 else \
        torch.randn((B, H, W, cin), generator=g, device='cuda') * 30.0
    w = (torch.rand((16, cin, 3, 3), generator=g, device='cuda') - 0.5) * 0.5
    b = torch.rand(16, generator=g, device='cuda') - 0.5
    xf = x.float() / 255.0 if dt == torch.uint8 else x
    ref = F.relu(F.conv2d(xf.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1))
    y = qops.conv3x3_thin(x, qops.pack_thin_weights(w), b, dtype=torch.float32)
    assert y.shape == ref.shape and y.dtype == torch.float32 and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
  buf = torch.zeros((B, 16, 48, 64), device='cuda').contiguous(memory_format=torch.channels_last)
  qops.conv3x3_thin(x, qops.pack_thin_weights(w), b, out=buf)
  assert torch.equal(buf[:, :, :H, :W], y) and float(buf[:, :, H:].abs().sum()) == 0 and float(buf[:, :, :, W:].abs().sum()) == 0
  w2 = (torch.rand((16, 16, 3, 3), generator=g, device='cuda') - 0.5) * 0.2
  b2 = torch.rand(16, generator=g, device='cuda') - 0.5
  pw = torch.rand(16, generator=g, device='cuda') - 0.5
  got = qops.conv3x3_relu_project(buf, qops.pack_conv3x3_weights_x3(w2), b2, pw, 0.25, H, W)
  act = F.relu(F.conv2d(buf.double()[:, :, :H, :W], w2.double(), b2.double(), padding=1))
  ref = (act * pw.double()[None, :, None, None]).sum(1) + 0.25
  assert got.shape == (B, H, W) and got.dtype == torch.float32
  assert float((got.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())


@pytest.mark.parametrize('cin,dt,H,W', [(2, torch.uint8, 128, 128), (1, torch.uint8, 32, 32), (2, torch.float32, 48, 16), (1, torch.float32, 16, 64)])
def test_fused_thin_and_second_layer_equal_the_two_kernels(cin, dt, H, W):
  """`thin_conv3x3_bias_relu` (one kernel, the 16-channel intermediate in LDS) against `conv3x3_thin` followed by
  `conv3x3_bias_relu` — each held to float64 by the tests above: the same arithmetic per value, so equal BIT FOR BIT, in every
  output form (new tensor; channel slice of a concat buffer + pooled tensor, the rest of the buffer untouched; channel-major)."""
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(31 + cin)
  B = 3
  x = torch.randint(0, 256, (B, H, W, cin), generator=g, device='cuda', dtype=torch.uint8) if dt == torch.uint8 else \
      torch.randn((B, H, W, cin), generator=g, device='cuda') * 3.0
  w1 = qops.pack_thin_weights((torch.rand((16, cin, 3, 3), generator=g, device='cuda') - 0.5) * 0.5)
  b1 = torch.rand(16, generator=g, device='cuda') - 0.5
  w2 = (torch.rand((16, 16, 3, 3), generator=g, device='cuda') - 0.5) * 0.2
  b2 = torch.rand(16, generator=g, device='cuda') - 0.5
  wf = qops.pack_conv3x3_weights_x3(w2)
  y = qops.conv3x3_thin(x, w1, b1, dtype=torch.float32)
  assert float(y.abs().max()) > 0
  ref = qops.conv3x3_bias_relu(y, wf, b2, 16)
  got = qops.thin_conv3x3_bias_relu(x, w1, b1, wf, b2)
  assert got.shape == ref.shape and got.dtype == torch.float32 and got.is_contiguous(memory_format=torch.channels_last)
  assert torch.equal(got, ref)
  cat_r = torch.full((B, 32, H, W), 7.0, device='cuda').contiguous(memory_format=torch.channels_last)
  cat_g = cat_r.clone(memory_format=torch.preserve_format)
  _, pr = qops.conv3x3_bias_relu(y, wf, b2, 16, out=cat_r, out_offset=16, pool=True)
  _, pg = qops.thin_conv3x3_bias_relu(x, w1, b1, wf, b2, out=cat_g, out_offset=16, pool=True)
  assert torch.equal(cat_g, cat_r) and torch.equal(pg, pr) and bool((cat_g[:, :16] == 7.0).all())
  assert torch.equal(pg, torch.nn.functional.max_pool2d(ref, 2))
  nr = qops.conv3x3_bias_relu(y, wf, b2, 16, nchw=True)
  ng = qops.thin_conv3x3_bias_relu(x, w1, b1, wf, b2, nchw=True)
  assert ng.is_contiguous() and torch.equal(ng, nr)


@pytest.mark.parametrize('H,W', [(97, 97), (40, 56), (5, 130), (16, 16)])
def test_fused_position_head_equals_the_two_kernels(H, W):
  """`pos_layers` as one kernel against the thin layer into a zero-margined map + the 16 -> 16 -> 1 tail: bit for bit, for
  maps that are not multiples of the 16 x 16 tile (the kernel's own border handling replaces the padded map)."""
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(41)
  B = 3
  x = torch.randn((B, H, W), generator=g, device='cuda') * 30.0
  w1 = qops.pack_thin_weights((torch.rand((16, 1, 3, 3), generator=g, device='cuda') - 0.5) * 0.5)
  b1 = torch.rand(16, generator=g, device='cuda') - 0.5
  w2 = (torch.rand((16, 16, 3, 3), generator=g, device='cuda') - 0.5) * 0.2
  b2 = torch.rand(16, generator=g, device='cuda') - 0.5
  pw = torch.rand(16, generator=g, device='cuda') - 0.5
  wf = qops.pack_conv3x3_weights_x3(w2)
  hp, wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
  buf = torch.zeros((B, 16, hp, wp), device='cuda').contiguous(memory_format=torch.channels_last)
  qops.conv3x3_thin(x.reshape(B, H, W, 1), w1, b1, out=buf)
  ref = qops.conv3x3_relu_project(buf, wf, b2, pw, -0.75, H, W)
  got = qops.thin_conv3x3_relu_project(x, w1, b1, wf, b2, pw, -0.75)
  assert got.shape == (B, H, W) and torch.equal(got, ref)


def test_fast_rollout_with_and_without_the_fused_first_level():
  """The fp32-class rollout forward with the fused kernels (default) and with the two-kernel forms: features and advantages
  equal bit for bit.  (Eight samples: a batch the deep levels' kernels take — with an odd batch those layers fall to the library's
  fp32 convolutions, whose results move by ~1e-5 of the scale from run to run.)"""
  from stackrl_amd import nets, qops
  net = nets.DeepQSiamFCN(seed=7).cuda().eval()
  g = torch.Generator(device='cuda').manual_seed(8)
  xm = torch.randint(0, 256, (8, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8)
  xo = torch.randint(0, 256, (8, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8)
  fa, fb = qops.FastFeatures(net, dtype=torch.float32), qops.FastFeatures(net, dtype=torch.float32, fuse_thin=False)
  (ax, aw), (bx, bw) = fa((xm, xo)), fb((xm, xo))
  assert torch.equal(ax, bx) and torch.equal(aw, bw)
  corr = torch.randn((8, 1, 97, 97), generator=g, device='cuda') * 20.0
  assert fa._pos is not None and torch.equal(fa.pos(corr), fb.pos(corr))


def test_fast_position_head_fp32_tracks_the_module():
  """`FastFeatures(dtype=float32).pos` against `net.pos` in fp32: fp32-class agreement (1e-4 of the advantage range)."""
  from stackrl_amd import nets, qops
  net = nets.DeepQSiamFCN(seed=9).cuda().eval()
  g = torch.Generator(device='cuda').manual_seed(2)
  corr = torch.randn((4, 1, 97, 97), generator=g, device='cuda') * 20.0
  ff = qops.FastFeatures(net, dtype=torch.float32)
  got = ff.pos(corr)
  assert ff._pos is not None     # the hand-written head ran, not the module
  with torch.no_grad():
    ref = net.pos(corr).flatten(1)
  assert got.shape == ref.shape == (4, 97 * 97) and got.dtype == torch.float32
  assert float((got - ref).abs().max()) <= 1e-4 * float(ref.max() - ref.min())


def test_fast_features_fp32_match_the_module():
  """The fp32 fast rollout (`FastFeatures(dtype=float32)`: bias-free library fp32 convolutions + the fused fp32 epilogues
  of csrc/epilogue.hip instead of separate bias / ReLU / max-pool / concatenate / layout kernels) against the stock fp32
  module: the same arithmetic up to where the bias is added, so fp32-level agreement (1e-5 of the feature scale), and the
  policy built on it picks the actions of `DQN.policy` at epsilon 0."""
  from stackrl_amd import nets, qops
  from stackrl_amd.dqn import DQN
  net = nets.DeepQSiamFCN(seed=5).cuda().eval()
  g = torch.Generator(device='cuda').manual_seed(4)
  xm = torch.randint(0, 256, (6, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8)
  xo = torch.randint(0, 256, (6, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8)
  with torch.no_grad():
    ex, _, ew = net.features((xm, xo))
  # (a) epilogues only: the same fp32 arithmetic up to where the bias is added
  fx, fw = qops.FastFeatures(net, dtype=torch.float32, x3_conv=False)((xm, xo))
  for got, ref in ((fx, ex), (fw, ew)):
    assert got.shape == ref.shape and got.dtype == torch.float32 and got.is_contiguous()
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
  # (b) + the 16- / 32-output-channel layers and the two upper transposed convolutions on the matrix cores as bf16x3
  #     products, the thin first layers on the vector ALU: fp32-class, 2e-4 of the feature scale after the whole U-Net
  #     (2^-16 per product, a dozen layers deep)
  gx, gw = qops.FastFeatures(net, dtype=torch.float32)((xm, xo))
  for got, ref in ((gx, ex), (gw, ew)):
    assert got.dtype == torch.float32 and float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
  agent = DQN(net, collect_batch_size=6, replay_memory_size=6 * 4, exploration=0.0)
  a_ref, q_ref = agent.policy((xm, xo), exploration=True, values=True)
  pol = qops.FusedPolicy(autocast=None, fast=True)
  a = pol(net, (xm, xo), 0.0, torch.Generator(device='cuda').manual_seed(1))
  # the same greedy actions, or actions whose Q-value is within the fp32-class tolerance of the maximum
  qa = q_ref.gather(1, a[:, None])[:, 0]
  assert bool((q_ref.amax(-1) - qa <= 2e-4 * q_ref.abs().amax(-1)).all())


@pytest.mark.parametrize('C,shape,relu', [(16, (3, 40, 56), True), (64, (2, 8, 8), True), (256, (32, 4, 4), True), (128, (5, 7, 3), False)])
def test_bias_act_autograd_matches_torch(C, shape, relu):
  """The fused bias + ReLU pass of the update path (forward in place, backward = masked gradient + bias gradient in a
  fixed summation order, csrc/epilogue.hip) against torch autograd in float64."""
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(C)
  B, H, W = shape
  x = torch.randn((B, C, H, W), generator=g, device='cuda').contiguous(memory_format=torch.channels_last)
  b = torch.randn(C, generator=g, device='cuda')
  gy = torch.randn((B, C, H, W), generator=g, device='cuda').contiguous(memory_format=torch.channels_last)
  xr, br = x.double().requires_grad_(True), b.double().requires_grad_(True)
  yr = xr + br[None, :, None, None]
  yr = torch.relu(yr) if relu else yr
  yr.backward(gy.double())
  xs = x.clone().requires_grad_(True)
  bs = b.clone().requires_grad_(True)
  y = qops.bias_act_autograd(xs * 1.0, bs, relu)     # `* 1.0`: a non-leaf, like a convolution's output
  assert float((y.double() - yr).abs().max()) <= 1e-6
  y.backward(gy)
  assert float((xs.grad.double() - xr.grad).abs().max()) == 0.0
  assert float((bs.grad.double() - br.grad).abs().max()) <= 2e-6 * max(1.0, float(br.grad.abs().max()))
  # deterministic: the same bits on a second evaluation
  xs2, bs2 = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
  qops.bias_act_autograd(xs2 * 1.0, bs2, relu).backward(gy)
  assert torch.equal(bs2.grad, bs.grad)


def test_fused_epilogue_net_matches_the_module_graph():
  """`DeepQSiamFCN.set_fused_epilogues` (the library update path, `DQN(hand_convs=False)`) against the plain module graph:
  the fused path changes no arithmetic, only the order of the bias-gradient sums.  Round 3 compared the two paths with
  each other at 1e-3 of a parameter's gradient scale, saw 1.33e-3 once on `left.down.3.0.weight` and widened the bound to 3e-3
  without knowing which side had moved.  Round 4 put a float64 evaluation on the host beside both and found the answer in the
  first run: the LIBRARY's float32 convolution gradients of the deep levels (64 ... 256 channels) sit 1.0 ... 1.4e-3 from
  float64 on BOTH paths, the same figure to two digits (`left.down.3.0.weight`: 1.33e-3 on either side, thousands of
  elements, every output channel) — the accuracy of the library's algorithm for those shapes, not noise and not corruption.
  The two paths normally share it (same kernels: their mutual difference is an order below); when the library picks a
  different algorithm for one of them the whole 1.3e-3 shows between them, which is what round 3 saw.  (The hand-written
  update path, the default, is held to float64 at 2e-3 and observed at 5e-5: tests/test_train_conv_gpu.py.)
  So each path is compared with float64, parameter by parameter, at 3e-3 — the library's measured accuracy class with a
  factor of two — and a larger deviation is accepted only in the shape a flipped ReLU mask gives it (below); the record of
  a failure names the side, the elements and the reference's smallest pre-activations."""
  import copy
  from stackrl_amd import nets
  net = nets.DeepQSiamFCN(seed=3).cuda()
  g = torch.Generator(device='cuda').manual_seed(8)
  xm = torch.randint(0, 256, (4, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8)
  xo = torch.randint(0, 256, (4, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8)
  gq = torch.randn((4, net.n_actions), generator=g, device='cuda')
  # float64 on the host, with the smallest |pre-activation| of every convolution recorded
  ref = copy.deepcopy(net).double().cpu()
  closest = {}
  for name, m in ref.named_modules():
    if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
      m.register_forward_hook(lambda mod, i, o, name=name: closest.__setitem__(name, float(o.detach().abs().min())))
  fx, fx0 = ref.left(xm.cpu().permute(0, 3, 1, 2).double() / 255.0)          # models.py:144-147 in float64
  fw, _ = ref.right(xo.cpu().permute(0, 3, 1, 2).double() / 255.0)
  qd = ref.head(ref.correlation(fx, fw), fx0)
  qd.backward(gq.double().cpu())
  want = {n: p.grad.cuda() for n, p in ref.named_parameters()}
  q0 = net((xm, xo)); q0.backward(gq)
  plain = {n: p.grad.clone() for n, p in net.named_parameters()}
  net.zero_grad(set_to_none=True)
  net.set_fused_epilogues(True)
  q1 = net((xm, xo)); q1.backward(gq)
  fused = {n: p.grad.clone() for n, p in net.named_parameters()}
  assert float((q1 - q0).abs().max()) <= 1e-5 * float(q0.abs().max())
  assert float((q0.double() - qd.cuda()).abs().max()) <= 1e-4 * float(qd.abs().max())
  # A deviation above 3e-3 is accepted only in the shape a flipped ReLU mask gives it (the float64 forward's smallest
  # |pre-activation| is 1e-8 ... 1e-7 with these inputs — below the rounding of a float32 convolution, whose library kernels
  # split the reduction and add atomically for some shapes, so the mask of such a pixel can differ between roundings of the
  # same forward): below 5e-2 of the scale and confined to at most two output channels of the layer.  Anything else — a
  # block of elements, several channels, a larger error — is corruption and fails with its location.
  bad, flips = [], []
  for n in want:
    scale = float(want[n].abs().max())
    if scale < 1e-9:
      continue
    for side, got in (('plain', plain[n]), ('fused', fused[n])):
      diff = (got.double() - want[n]).abs()
      e = float(diff.max()) / scale
      if e <= 3e-3:
        continue
      d = (diff > 1e-4 * scale).nonzero()
      rng = [(int(d[:, k].min()), int(d[:, k].max())) for k in range(d.shape[1])]
      oc_dim = 1 if ('.up.' in n and n.endswith('weight')) else 0          # ConvTranspose2d weights are [cin, cout, 2, 2]
      channels = sorted(set(d[:, oc_dim].tolist()))
      rec = '{} [{}]: {:.2e} of scale {:.3g}; {} elements off by > 1e-4 in output channels {}, index ranges {}'.format(
        n, side, e, scale, d.shape[0], channels[:8], rng)
      (flips if (e <= 5e-2 and len(channels) <= 2) else bad).append(rec)
  near = sorted(closest.items(), key=lambda kv: kv[1])[:4]
  if flips:
    print('accepted as ReLU-mask flips:', '; '.join(flips), '| smallest |pre-activation| of the float64 forward:', near)
  assert not bad, '; '.join(bad) + ' | smallest |pre-activation| of the float64 forward: {}'.format(near)


def test_update_stays_finite_under_the_concurrent_env_step():
  """configs[4]'s per-GPU shard (2,048 envs x 32 rocks, 64^2 maps), the loop of `Training.run`: the graph-replayed update
  runs while the env step occupies the GPU from its side stream.  With the library's bias-gradient reductions this
  ended in a corrupt gradient element within ten updates in 5 runs of 6 (update-path kernels);
  the fused bias / ReLU passes must keep the gradient bucket, the parameters and the Adam state finite and small."""
  from stackrl_amd import assets, env as envs, nets, qops
  from stackrl_amd.dqn import DQN, PolynomialDecay
  from stackrl_amd.training import Trainer
  B, L = 2048, 32
  env = envs.make('Stack-v0', n_parallel=B, seed=11, pool=assets.default_pool(), episode_length=L, side_stream=True,
                  resolution_factor=4)
  net = nets.DeepQSiamFCN(env.observation_spec, seed=1).cuda()
  agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=32, replay_memory_size=B * 16,
              discount_factor=.966667, collect_batch_size=B, exploration=PolynomialDecay(1.0, 400000, .1), prioritization=0.6,
              priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=7,
              policy_op=qops.FusedPolicy(autocast=torch.bfloat16, fast=True), xcorr='bf16x3', graphs=True)
  assert net.fused_epilogues
  tr = Trainer(env, agent)
  tr.initialize(num_steps=4)
  step = env.reset()
  agent.acknowledge_reset()
  for it in range(16):
    if callable(step):
      step = step()
    action = agent.collect(*step)
    step = env.step(action)            # side stream: runs underneath the update
    loss, _ = agent.train()
    g = agent._flat_grad
    assert bool(torch.isfinite(g).all()) and float(g.abs().max()) < 1e4, 'update {}: gradient bucket corrupt'.format(it)
    assert math.isfinite(float(loss))
  step() if callable(step) else None
  opt = agent._optimizer
  assert agent._train_graph is not None
  assert bool(torch.isfinite(opt.flat).all()) and bool(torch.isfinite(opt.m).all()) and bool(torch.isfinite(opt.v).all())
  env.close()


_GEMM_LAYERS = [(32, 64, 32), (64, 64, 32), (128, 64, 32), (64, 128, 16), (128, 128, 16), (256, 128, 16), (128, 256, 8), (256, 256, 8),
                (32, 64, 8), (64, 64, 16), (128, 64, 16), (256, 128, 8), (128, 256, 4)]


@pytest.mark.parametrize('cin,cout,W', _GEMM_LAYERS)
@pytest.mark.parametrize('f32', [False, True])
def test_conv3x3_gemm_matches_torch_fp64(cin, cout, W, f32):
  """The implicit-GEMM convolution of the deep U-Net levels (`k_conv3x3_gemm`) + bias + ReLU against float64 torch on the
  same (bf16-rounded, for the bf16 form) operands: bf16 output rounding (2^-8) / the bf16x3 split's 3e-5 of the output scale;
  plain output and a channel slice of a wider buffer."""
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(cin + cout + W)
  B = 8
  dt = torch.float32 if f32 else torch.bfloat16
  x = torch.randn(B, cin, W, W, generator=g, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
  w = torch.randn(cout, cin, 3, 3, generator=g, device='cuda') / (3 * cin ** 0.5)
  b = torch.randn(cout, generator=g, device='cuda') * 0.1
  wr = w if f32 else w.to(torch.bfloat16).float()
  ref = torch.relu(torch.nn.functional.conv2d(x.double(), wr.double(), b.double(), padding=1))
  assert qops.conv3x3_gemm_supported(cin, cout, W, B)
  wf = qops.pack_conv3x3_gemm_weights(w, x3=f32)
  y = qops.conv3x3_gemm_bias_relu(x, wf, b, cout)
  scale = float(ref.abs().max())
  tol = 3e-5 * scale if f32 else None
  err = (y.double() - ref).abs()
  if f32:
    assert y.dtype == torch.float32 and float(err.max()) <= tol
  else:
    assert y.dtype == torch.bfloat16 and bool((err <= 2.0 ** -8 * ref.abs().clamp(min=1e-2 * scale)).all())
  cat = torch.full((B, 2 * cout, W, W), 3.0, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last)
  qops.conv3x3_gemm_bias_relu(x, wf, b, cout, out=cat, out_offset=cout)
  assert torch.equal(cat[:, cout:], y) and bool((cat[:, :cout] == 3.0).all())


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_pool2x2_of_a_channel_slice(dt):
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(3)
  buf = torch.randn((3, 48, 12, 20), generator=g, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
  got = qops.pool2x2(buf, 32, 16)
  assert got.is_contiguous(memory_format=torch.channels_last)
  assert torch.equal(got, torch.nn.functional.max_pool2d(buf[:, 16:].float(), 2).to(dt))


@pytest.mark.parametrize('cin,cout,H,W', [(128, 64, 16, 16), (256, 128, 8, 8), (128, 64, 5, 7)])
@pytest.mark.parametrize('f32', [False, True])
def test_convt2x2_gemm_matches_torch_fp64(cin, cout, H, W, f32):
  """The transposed convolutions of the deep levels as a GEMM (`k_convt2x2_gemm`) + bias + ReLU into a slice of the concat
  buffer against float64 torch, both precisions; an odd map size exercises the partial last pixel tile."""
  from stackrl_amd import qops
  F = torch.nn.functional
  g = torch.Generator(device='cuda').manual_seed(cin + H)
  B = 3
  dt = torch.float32 if f32 else torch.bfloat16
  x = torch.randn((B, cin, H, W), generator=g, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
  w = torch.randn((cin, cout, 2, 2), generator=g, device='cuda') / cin ** 0.5
  b = torch.randn(cout, generator=g, device='cuda') * 0.1
  wr = w if f32 else w.to(torch.bfloat16).float()
  ref = F.relu(F.conv_transpose2d(x.double(), wr.double(), b.double(), stride=2))
  cat = torch.full((B, 2 * cout, 2 * H, 2 * W), 3.0, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last)
  qops.convt2x2_gemm_bias_relu(x, qops.pack_convt2x2_weights(w, x3=f32), b, cout, cat, 0)
  err = (cat[:, :cout].double() - ref).abs()
  if f32:
    assert float(err.max()) <= 3e-5 * float(ref.abs().max())
  else:
    assert bool((err <= 2.0 ** -8 * ref.abs().clamp(min=1e-2 * float(ref.abs().max()))).all())
  assert bool((cat[:, cout:] == 3.0).all())


def test_prefetch_fifo_replays_from_the_update_graph(ref_pool):
  """`DQN(prefetch=2)` (the reference's `dataset.prefetch`, dqn.py:247-252): the FIFO of minibatches lives at fixed
  addresses, so the hipGraph-replayed update must use, update for update, the same minibatches as the eager one and land on
  the same weights (the update's kernels are deterministic); and the hand-written update (`HandNet`) against the module
  graph through the library within the float32-class tolerance of the cross-correlation."""
  from stackrl_amd import env as envs, nets, qops
  from stackrl_amd.dqn import DQN, PolynomialDecay
  from stackrl_amd.training import Trainer
  B, L = 8, 4
  runs = []
  for kw in (dict(graphs=False), dict(graphs=True), dict(graphs=False, hand_convs=False)):
    env = envs.make('Stack-v0', n_parallel=B, seed=5, pool=ref_pool, episode_length=L)
    net = nets.DeepQSiamFCN(env.observation_spec, seed=2).cuda()
    agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=8, replay_memory_size=B * 16,
                discount_factor=.966667, collect_batch_size=B, exploration=1.0, prioritization=0.6,
                priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=9,
                policy_op=qops.FusedPolicy(), xcorr='bf16x3', prefetch=2, **kw)
    assert (agent._hand is not None) == kw.get('hand_convs', True)
    tr = Trainer(env, agent)
    tr.initialize(num_steps=10)
    used, losses = [], []
    for _ in range(7):                     # 3 eager warm-up updates, the capture, 3 replays (graphs=True)
      loss, _ = agent.train()
      used.append(agent._last_sample_indexes.clone()); losses.append(float(loss))
    runs.append((used, losses, [p.detach().clone() for p in net.parameters()], agent._train_graph is not None))
    env.close()
  (ua, la, wa, ga), (ub, lb, wb, gb), (uc, lc, wc, gc) = runs
  assert not ga and gb and not gc
  # the graph's own `indexes` tensor is static: what it held after each replay is what that update used
  for x, y in zip(ua, ub):
    assert torch.equal(x, y)
  assert la == lb
  for p, q in zip(wa, wb):
    assert torch.equal(p, q)
  for x, y in zip(ua[:1], uc[:1]):         # the library path draws the same first minibatch; later priorities differ in the last bits
    assert torch.equal(x, y)
  assert abs(la[0] - lc[0]) <= 1e-3 * max(1e-6, abs(lc[0]))


def test_groupwise_collection_equals_one_collect(ref_pool):
  """`PipelinedVecStackEnv` + `Trainer.collect_step`: the batch as two handles that step as their actions arrive, the policy
  evaluated group by group with the random numbers of one `collect` over the batch.  Same seeds -> the same trajectories,
  replay contents and weights as the one-handle loop, bit for bit (every kernel of the rollout works sample by sample once
  a group holds >= 256 samples; the update's kernels are deterministic)."""
  from stackrl_amd import env as envs, nets, qops
  from stackrl_amd.dqn import DQN, PolynomialDecay
  from stackrl_amd.training import Trainer
  B, L = 512, 3
  runs = []
  for groups in (None, 2):
    env = envs.make('Stack-v0', n_parallel=B, seed=5, pool=ref_pool, episode_length=L, side_stream=True,
                    **({} if groups is None else dict(groups=groups)))
    assert getattr(env, 'groups', 1) == (groups or 1)
    net = nets.DeepQSiamFCN(env.observation_spec, seed=2).cuda()
    agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=8, replay_memory_size=B * 8,
                discount_factor=.966667, collect_batch_size=B, exploration=0.5, prioritization=0.6,
                priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=9,
                policy_op=qops.FusedPolicy(chunk=256, fast=True), xcorr='bf16x3', prefetch=3)
    tr = Trainer(env, agent)
    tr.initialize(num_steps=2)
    losses = tr.run(2 * (L + 1) + 1)       # two episodes and a step: through the auto-reset call of every group
    mem = agent._replay_memory
    runs.append((losses.clone(), [p.detach().clone() for p in net.parameters()], mem._actions.clone(), mem._rewards.clone(),
                 mem._states[0].clone(), mem._states[1].clone(), float(tr.returns) if tr.returns is not None else None))
    env.close()
  a, b = runs
  assert torch.equal(a[2], b[2]), 'actions'
  assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]) and torch.equal(a[5], b[5]), 'stored transitions'
  assert torch.equal(a[0], b[0]), 'losses'
  for p, q in zip(a[1], b[1]):
    assert torch.equal(p, q)
  assert a[6] == b[6]


@pytest.mark.parametrize('graphs', [False, True])
def test_early_gradient_equals_the_serial_update(ref_pool, graphs):
  """`DQN(early_gradient=True)`: the gradient half of an update (target evaluation, forward, loss, backward on the minibatch
  at the head of the prefetch FIFO) runs on its own stream beside the collect step, the rest (new minibatch into the FIFO,
  optimiser step, priorities) after it.  Same operands, same operations: losses, sampled indexes, priorities and weights are
  those of the serial update, bit for bit — eager and replayed from hipGraphs, through a target sync."""
  from stackrl_amd import env as envs, nets, qops
  from stackrl_amd.dqn import DQN, PolynomialDecay
  from stackrl_amd.training import Trainer
  B, L = 64, 3
  runs = []
  for early in (False, True):
    env = envs.make('Stack-v0', n_parallel=B, seed=5, pool=ref_pool, episode_length=L, side_stream=True)
    net = nets.DeepQSiamFCN(env.observation_spec, seed=2).cuda()
    agent = DQN(net, learning_rate=6.25e-5, adam_betas=(0.95, 0.95), minibatch_size=8, replay_memory_size=B * 8,
                discount_factor=.966667, collect_batch_size=B, exploration=0.5, prioritization=0.6, target_update_period=4,
                priority_bias_compensation=PolynomialDecay(0.4, 400000, 1.0), double=True, seed=9,
                policy_op=qops.FusedPolicy(fast=True), xcorr='bf16x3', prefetch=2, graphs=graphs, early_gradient=early)
    assert agent._early == early
    tr = Trainer(env, agent)
    tr.initialize(num_steps=2)
    losses = tr.run(11)
    started = agent._grad_graph is not None if graphs else None
    mem = agent._replay_memory
    runs.append((losses.clone(), [p.detach().clone() for p in net.parameters()], [p.detach().clone() for p in agent._target_q_net.parameters()],
                 mem._logits.clone(), mem._actions.clone(), agent._last_sample_indexes.clone(), started))
    env.close()
  a, b = runs
  assert b[6] in (None, True), 'the early form of the graph was never captured'
  assert torch.equal(a[4], b[4]), 'actions'
  assert torch.equal(a[0], b[0]), 'losses {} vs {}'.format(a[0].tolist(), b[0].tolist())
  assert torch.equal(a[5], b[5]) and torch.equal(a[3], b[3]), 'sampled indexes / priorities'
  for p, q in zip(a[1] + a[2], b[1] + b[2]):
    assert torch.equal(p, q)


@pytest.mark.parametrize('n', [5, 1024, 65536, 300001])
def test_logit_extrema_matches_torch(n):
  """srl_logit_extrema (the replay memory's tracker scans, memory.py:164-177, :282-316): max logit and min FINITE logit with
  their lowest indexes, against torch — with unsampleable rows (-inf), ties, and a memory without any finite row."""
  from stackrl_amd import qops
  g = torch.Generator(device='cuda').manual_seed(n)
  x = torch.randn(n, generator=g, device='cuda')
  x[torch.rand(n, generator=g, device='cuda') < 0.3] = -math.inf
  x[n // 2] = x[n - 1] = 7.5                     # a tie at the maximum
  x[1] = x[n // 3] = -9.25                       # a tie at the minimum
  ws = {}
  for t in (x, torch.full((n,), -math.inf, device='cuda')):
    (mv, mi), (nv, ni) = qops.logit_extrema(t, ws)
    assert float(mv) == float(t.max()) and int(mi) == int((t == t.max()).nonzero()[0, 0])
    fin = torch.isfinite(t)
    if bool(fin.any()):
      ref = t[fin].min()
      assert float(nv) == float(ref) and int(ni) == int((t == ref).nonzero()[0, 0])
    else:
      assert float(nv) == math.inf and int(ni) == 0


def test_rollout_argmax_against_float64_on_real_observations():
  """The rollout's precision against the reference's, at the level that matters (dqn.py:330-348: the greedy action is the
  arg-max over the 9,409 placements; the reference computes in float32, models.py:144-147).  4,608 real observations
  (random-policy episodes), a He-initialised net, advantages in float64 as the yardstick: the fp32-class rollout
  (`dqn.bf16x3` of the bench line) picks float64's action wherever float64's top-2 gap exceeds 1e-4 of the advantage range,
  stays within 1e-4 of the range everywhere, and — the comparison the tolerance is judged by — does not flip more often than
  a few times what the stock FLOAT32 module (the reference's own dtype, library convolutions) flips.  The numbers of a run
  are in profiles/r05_rollout_argmax.json (tools/argmax_precision.py)."""
  import importlib.util, os
  spec = importlib.util.spec_from_file_location('argmax_precision', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'argmax_precision.py'))
  ap = importlib.util.module_from_spec(spec); spec.loader.exec_module(ap)
  r = ap.run(4608, seed=1)
  print(r)
  assert r['samples'] == 4608 and r['actions'] == 9409
  x3, f32, b16 = r['bf16x3'], r['f32'], r['bf16']
  assert x3['max_rel_err'] <= 1e-4, x3                       # advantages within 1e-4 of their range of float64's
  assert x3['largest_gap_with_a_flip'] <= 1e-4, x3           # arg-max-exact above a top-2 gap of 1e-4 of the range
  assert x3['max_regret_of_a_flip'] <= 1e-4, x3              # and a flip gives up at most that much advantage
  assert x3['flip_rate'] <= max(5 * f32['flip_rate'], 0.01), (x3, f32)
  assert b16['max_rel_err'] > x3['max_rel_err']              # (the labelled bf16 rollout is the narrower one)


@pytest.mark.parametrize('B,C,dt', [(3, 16, 'f32x3'), (2, 7, 'f32x3'), (2, 16, 'f32'), (3, 16, 'bf16'), (200, 16, 'f32x3')])
def test_xcorr_row_product_forward(B, C, dt, monkeypatch):
  """The rollout's cross-correlation forward as a product per map row (`k_xcorr_rows`, csrc/xcorr_mfma.hip: Hankel fragments of
  the row x the kernel's rows, the sum over kernel rows along a diagonal by DPP lane shifts): against the library formulation
  in float64 at the kernel family's stated tolerances, against the Toeplitz kernel it replaces for large batches, bit-identical
  on repetition, and chosen by batch size (>= 192 samples) when nothing forces it."""
  from stackrl_amd import nets, qops
  g = torch.Generator(device='cuda').manual_seed(B * 17 + C)
  x = torch.rand((B, C, 128, 128), generator=g, device='cuda')
  w = torch.rand((B, C, 32, 32), generator=g, device='cuda') - 0.3
  if dt == 'bf16':
    x, w = x.to(torch.bfloat16), w.to(torch.bfloat16)
  prec = {'f32x3': qops.BF16X3, 'f32': qops.BF16, 'bf16': qops.BF16}[dt]
  def run():
    return qops.xcorr_forward_mfma(x, w, prec)
  if B < 192:
    monkeypatch.setenv('SRL_XCORR_ROWS', '1')
  rows = run()
  assert torch.equal(rows, run())
  ref = nets.correlation_reference(x.double(), w.double())
  scale = float(ref.abs().max())
  tol = 2e-5 if dt != 'f32' else 6e-3          # operands split (fp32-class) or exact bf16 operands / operands rounded to bf16
  assert rows.shape == ref.shape and rows.dtype == torch.float32
  assert float((rows.double() - ref).abs().max()) <= tol * scale
  monkeypatch.setenv('SRL_XCORR_ROWS', '0')
  toep = run()
  assert float((rows - toep).abs().max()) <= 2e-5 * scale      # the same products, another order of the fp32 sums
