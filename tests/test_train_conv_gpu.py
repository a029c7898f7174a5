"""GPU tests of the update path's hand-written convolutions (csrc/train_conv.hip, stackrl_amd/qtrain.py) against float64
torch: forward, data gradient and weight gradient of the 3 x 3 layers, the 2 x 2 transposed layers, the activation-gradient
pass with the max-pool routing, and the whole `DeepQSiamFCN` forward / backward against the module graph's autograd.

The kernels multiply and accumulate in float32 (v_mfma_f32_16x16x4_f32): the stated tolerance against float64 is 2e-5 of
each tensor's scale (accumulation order over up to 2,304 x pixels terms); the whole net goes through the bf16x3
cross-correlation (2e-5 per application, stated in tests/test_learner_gpu.py), so its tolerance is 2e-4."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(autouse=True)
def _poisoned_scratch():
  """Every scratch buffer the kernels of this file are handed is filled with NaN first (`qtrain._Scratch.poison`): a
  partial-sum slot that a finishing kernel reads without anybody having written it shows as NaN instead of passing on
  whatever the allocator left there (round 3 had one unexplained weight-gradient mismatch)."""
  from stackrl_amd import qtrain
  old = qtrain._Scratch.poison
  qtrain._Scratch.poison = True
  yield
  qtrain._Scratch.poison = old


def _rel(got, want):
  got, want = got.detach(), want.detach().double()
  return float((got.double() - want).abs().max()) / max(float(want.abs().max()), 1e-30)


def _where(got, want):
  """Where a tensor leaves its reference (for the assertion message: a fault that shows once is localised by its record)."""
  got, want = got.detach().double(), want.detach().double()
  bad = ((got - want).abs() > 1e-3 * want.abs().max()).nonzero()
  if bad.numel() == 0:
    return 'no element off by more than 1e-3 of the scale'
  rng = [(int(bad[:, d].min()), int(bad[:, d].max())) for d in range(bad.shape[1])]
  return '{} elements off; index ranges per dim {}; first {}; got {} want {}'.format(
    bad.shape[0], rng, bad[0].tolist(), float(got[tuple(bad[0])]), float(want[tuple(bad[0])]))


def _packed(mods):
  from stackrl_amd import qtrain
  net = torch.nn.Sequential(*mods).cuda()
  P = qtrain.Packed(net)
  P.refresh()
  return net, P


@pytest.mark.parametrize('cin,cout,B,H,W', [(2, 16, 2, 32, 48), (1, 16, 3, 97, 97), (16, 16, 2, 48, 32), (32, 64, 5, 16, 16),
                                            (64, 32, 6, 8, 8), (128, 64, 3, 4, 4), (256, 256, 4, 8, 8), (48, 16, 1, 20, 17)])
def test_conv3x3_forward_data_and_weight_gradients_match_torch_fp64(cin, cout, B, H, W):
  from stackrl_amd import qtrain
  F = torch.nn.functional
  g = torch.Generator(device='cuda').manual_seed(cin * 7 + cout + H)
  (conv,), P = _packed([torch.nn.Conv2d(cin, cout, 3, padding=1)])
  with torch.no_grad():
    conv.bias.copy_(torch.randn(cout, generator=g, device='cuda') * 0.1)
  # the input as a channel slice of a wider buffer, the output into a channel slice of another
  xbuf = torch.randn((B, H, W, cin + 4), generator=g, device='cuda')
  x = qtrain.Act(xbuf, cin, 4 if cin % 4 == 0 else 0) if cin % 4 == 0 else qtrain.Act(xbuf[..., :cin].contiguous())
  xd = x.dense().permute(0, 3, 1, 2).double().cpu()
  ybuf = torch.full((B, H, W, cout + 8), 7.0, device='cuda')
  y = qtrain.tconv(x, P.w(conv, 0), conv.bias, cout, relu=True, out=(ybuf, 8))
  ref = F.relu(F.conv2d(xd, conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), padding=1)).cuda()
  assert _rel(y.dense().permute(0, 3, 1, 2), ref) <= TOL
  assert bool((ybuf[..., :8] == 7.0).all())
  # backward: activation gradient (ReLU mask + bias gradient), weight gradient, data gradient
  gy = torch.randn((B, H, W, cout), generator=g, device='cuda')
  sc = qtrain._Scratch()
  gb = torch.zeros(cout, device='cuda')
  gz = qtrain.tact_bwd(qtrain.Act(gy), y, sc, gbias=gb, relu=True)
  assert torch.equal(gz, gy * (y.dense() > 0))
  gw = torch.zeros_like(conv.weight)
  qtrain.twrw(x, gz, gw, sc)
  # reference gradients: float64 on the HOST (the framework's CPU convolution — no GPU library on the reference side); the
  # device library's float64 result rides along only to say, in a failure's message, which side left the other two
  def grads(dev):
    wd_ = conv.weight.detach().double().to(dev).requires_grad_(); bd_ = conv.bias.detach().double().to(dev).requires_grad_()
    xd_ = x.dense().permute(0, 3, 1, 2).double().to(dev).requires_grad_()
    F.relu(F.conv2d(xd_, wd_, bd_, padding=1)).backward(gy.permute(0, 3, 1, 2).double().to(dev))
    return wd_.grad, bd_.grad, xd_.grad
  wg, bg, xg = (t.cuda() for t in grads('cpu'))

  def third_opinion():
    lw, lb, lx = grads('cuda')
    return ' [device-library float64 against the host reference: weight {:.1e}, bias {:.1e}, data {:.1e}]'.format(
      _rel(lw, wg), _rel(lb, bg), _rel(lx, xg))
  assert bool(torch.isfinite(gw).all()), 'weight gradient holds a poisoned (never written) partial sum: ' + _where(gw, wg)
  if _rel(gw, wg) > TOL:
    raise AssertionError('weight gradient: ' + _where(gw, wg) + third_opinion())
  if _rel(gb, bg) > TOL:
    raise AssertionError('bias gradient: ' + _where(gb, bg) + third_opinion())
  cpad = (cin + 15) // 16 * 16
  gx = qtrain.tconv(qtrain.Act(gz), P.w(conv, 1), None, cpad, relu=False)
  if _rel(gx.t[..., :cin].permute(0, 3, 1, 2), xg) > TOL:
    raise AssertionError('data gradient: ' + _where(gx.t[..., :cin].permute(0, 3, 1, 2), xg) + third_opinion())
  if cpad != cin:
    assert float(gx.t[..., cin:].abs().max()) == 0.0
  # bit-identical on repetition (fixed-order reductions, no atomics)
  gw2 = torch.zeros_like(gw); qtrain.twrw(x, gz, gw2, sc)
  assert torch.equal(gw, gw2)


@pytest.mark.parametrize('cin,cout,B,H,W', [(32, 16, 2, 24, 16), (64, 32, 3, 8, 8), (256, 128, 4, 8, 8), (128, 64, 2, 5, 7)])
def test_transposed_conv_forward_and_gradients_match_torch_fp64(cin, cout, B, H, W):
  """`up{i}` (layers.py:222-229) as a 1 x 1 convolution to 4 cout channels + depth-to-space, into the first half of a
  concatenation buffer; its gradients from the space-to-depth activation gradient."""
  from stackrl_amd import qtrain
  F = torch.nn.functional
  g = torch.Generator(device='cuda').manual_seed(cin + cout + H)
  (up,), P = _packed([torch.nn.ConvTranspose2d(cin, cout, 2, stride=2)])
  with torch.no_grad():
    up.bias.copy_(torch.randn(cout, generator=g, device='cuda') * 0.1)
  x = qtrain.Act(torch.randn((B, H, W, cin), generator=g, device='cuda'))
  cat = torch.full((B, 2 * H, 2 * W, 2 * cout), 3.0, device='cuda')
  y = qtrain.tconv(x, P.w(up, 2), up.bias, 4 * cout, taps=1, relu=True, out=(cat, 0), d2s=cout)
  # float64 reference on the host (no GPU library on the reference side)
  xd = x.t.permute(0, 3, 1, 2).double().cpu().requires_grad_()
  wd = up.weight.detach().double().cpu().requires_grad_(); bd = up.bias.detach().double().cpu().requires_grad_()
  ref = F.relu(F.conv_transpose2d(xd, wd, bd, stride=2))
  assert _rel(y.dense().permute(0, 3, 1, 2), ref.cuda()) <= TOL
  assert bool((cat[..., cout:] == 3.0).all())
  gcat = torch.randn((B, 2 * H, 2 * W, 2 * cout), generator=g, device='cuda')
  ref.backward(gcat[..., :cout].permute(0, 3, 1, 2).double().cpu())
  sc = qtrain._Scratch()
  gb = torch.zeros(cout, device='cuda')
  gz = qtrain.tact_bwd(qtrain.Act(gcat, cout, 0), y, sc, gbias=gb, relu=True, s2d=True)
  assert tuple(gz.shape) == (B, H, W, 4 * cout)
  gw = torch.zeros_like(up.weight)
  qtrain.twrw(x, gz, gw, sc, taps=1, convt=True)
  gx = qtrain.tconv(qtrain.Act(gz), P.w(up, 3), None, cin, taps=1, relu=False)
  assert bool(torch.isfinite(gw).all()) and bool(torch.isfinite(gb).all())
  assert _rel(gb, bd.grad.cuda()) <= TOL, 'bias gradient: ' + _where(gb, bd.grad.cuda())
  assert _rel(gw, wd.grad.cuda()) <= TOL, 'weight gradient: ' + _where(gw, wd.grad.cuda())
  assert _rel(gx.t.permute(0, 3, 1, 2), xd.grad.cuda()) <= TOL, 'data gradient: ' + _where(gx.t.permute(0, 3, 1, 2), xd.grad.cuda())


@pytest.mark.parametrize('C,B,H,W', [(16, 2, 16, 24), (64, 3, 8, 8), (256, 2, 4, 4)])
def test_activation_gradient_routes_the_max_pool_like_the_library(C, B, H, W):
  """gz = (g + pool gradient to the first maximum of each 2 x 2 window) * [y > 0] against autograd of
  relu -> (identity, max_pool2d); y holds ties (zeros after the ReLU and equal positive values)."""
  from stackrl_amd import qtrain
  F = torch.nn.functional
  gen = torch.Generator(device='cuda').manual_seed(C + H)
  pre = torch.randn((B, H, W, C), generator=gen, device='cuda').round(decimals=0)      # integers: many exact ties
  ybuf = torch.zeros((B, H, W, 2 * C), device='cuda')
  ybuf[..., C:] = F.relu(pre)
  y = qtrain.Act(ybuf, C, C)
  g = torch.randn((B, H, W, 2 * C), generator=gen, device='cuda')
  gp = torch.randn((B, H // 2, W // 2, C), generator=gen, device='cuda')
  pd = pre.permute(0, 3, 1, 2).double().requires_grad_()
  yd = F.relu(pd)
  (yd * g[..., C:].permute(0, 3, 1, 2).double()).sum().backward(retain_graph=True)
  (F.max_pool2d(yd, 2) * gp.permute(0, 3, 1, 2).double()).sum().backward()
  sc = qtrain._Scratch()
  gb = torch.zeros(C, device='cuda')
  gz = qtrain.tact_bwd(qtrain.Act(g, C, C), y, sc, gbias=gb, gpool=gp, relu=True)
  want = pd.grad.permute(0, 2, 3, 1)
  assert float((gz.double() - want).abs().max()) <= 1e-6
  assert _rel(gb, want.sum(dim=(0, 1, 2))) <= 1e-5


@pytest.mark.parametrize('B,n,P,C,U', [(6, 4, 64, 256, 256), (3, 3, 16, 128, 40), (2, 1, 1, 48, 300)])
def test_value_branch_forward_and_gradients_match_torch_fp64(B, n, P, C, U):
  """`layers.value` (layers.py:424-436): average pool -> Dense + ReLU -> Dense(1), forward and — for the first n samples — the
  gradients of both dense layers and of the bottom features (added to an incoming gradient), against float64 on the host."""
  from stackrl_amd import qtrain, qops
  L = qtrain._lib()
  g = torch.Generator(device='cuda').manual_seed(B + C)
  side = int(round(P ** 0.5))
  x0 = torch.randn((B, side, side, C), generator=g, device='cuda')
  d1, d2 = torch.nn.Linear(C, U).cuda(), torch.nn.Linear(U, 1).cuda()
  pooled, hid = torch.empty((B, C), device='cuda'), torch.empty((B, U), device='cuda')
  v = torch.empty(B, device='cuda')
  st = qops._stream(v)
  assert L.srl_tvalue_fwd(x0.data_ptr(), d1.weight.data_ptr(), d1.bias.data_ptr(), d2.weight.data_ptr(), d2.bias.data_ptr(),
                          pooled.data_ptr(), hid.data_ptr(), v.data_ptr(), B, P, C, U, st) == 0
  xd = x0.double().cpu().requires_grad_()
  w1, b1 = d1.weight.detach().double().cpu().requires_grad_(), d1.bias.detach().double().cpu().requires_grad_()
  w2, b2 = d2.weight.detach().double().cpu().requires_grad_(), d2.bias.detach().double().cpu().requires_grad_()
  vr = (torch.relu(xd.mean(dim=(1, 2)) @ w1.T + b1) @ w2.T + b2)[:, 0]
  assert _rel(v, vr.cuda()) <= 1e-5
  gv = torch.randn(n, generator=g, device='cuda')
  gin = torch.randn((n, side, side, C), generator=g, device='cuda')
  (vr[:n] * gv.double().cpu()).sum().backward()
  gx = torch.full((n, side, side, C), float('nan'), device='cuda')
  gw1, gb1 = torch.full_like(d1.weight, float('nan')), torch.full_like(d1.bias, float('nan'))
  gw2, gb2 = torch.full_like(d2.weight, float('nan')), torch.full_like(d2.bias, float('nan'))
  sc = torch.full((n * U,), float('nan'), device='cuda')
  assert L.srl_tvalue_bwd(gv.data_ptr(), hid.data_ptr(), pooled.data_ptr(), d1.weight.data_ptr(), d2.weight.data_ptr(), gin.data_ptr(),
                          gx.data_ptr(), gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), sc.data_ptr(), n, P, C, U, st) == 0
  assert _rel(gw1, w1.grad.cuda()) <= 1e-5 and _rel(gb1, b1.grad.cuda()) <= 1e-5
  assert _rel(gw2, w2.grad.cuda()) <= 1e-5 and _rel(gb2, b2.grad.cuda()) <= 1e-5
  assert _rel(gx, (gin.double().cpu() + xd.grad[:n]).cuda()) <= 1e-6
  # bit-identical on repetition
  gw1b = torch.empty_like(gw1)
  assert L.srl_tvalue_bwd(gv.data_ptr(), hid.data_ptr(), pooled.data_ptr(), d1.weight.data_ptr(), d2.weight.data_ptr(), gin.data_ptr(),
                          gx.data_ptr(), gw1b.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), sc.data_ptr(), n, P, C, U, st) == 0
  assert torch.equal(gw1, gw1b)


def test_layout_passes_are_exact():
  """The copies around the cross-correlation (channels-last <-> channel-major, channel 0 of a gradient plain and zero-padded,
  flipped kernels) and the uint8 -> float32 / 255 input scaling, against the framework formulations, bit for bit."""
  from stackrl_amd import qtrain, qops
  L = qtrain._lib()
  g = torch.Generator(device='cuda').manual_seed(4)
  buf = torch.randn((3, 9, 7, 24), generator=g, device='cuda')
  a = qtrain.Act(buf, 16, 4)
  n = qtrain.to_nchw(a)
  assert torch.equal(n, buf[..., 4:20].permute(0, 3, 1, 2))
  assert torch.equal(qtrain.to_nhwc(n).t, buf[..., 4:20].contiguous())
  gr = torch.randn((2, 11, 11, 16), generator=g, device='cuda')
  plain, padded = torch.full((2, 11, 11), 5.0, device='cuda'), torch.full((2, 17, 17), 5.0, device='cuda')
  assert L.srl_tcorr_grad(gr.data_ptr(), 16, plain.data_ptr(), padded.data_ptr(), 2, 11, 3, qops._stream(gr)) == 0
  assert torch.equal(plain, gr[..., 0]) and torch.equal(padded, torch.nn.functional.pad(gr[..., 0], (3, 3, 3, 3)))
  w = torch.randn((5, 4, 6, 6), generator=g, device='cuda')
  out = torch.empty((3, 4, 6, 6), device='cuda')
  assert L.srl_tflip(w.data_ptr(), out.data_ptr(), 3 * 4, 36, qops._stream(w)) == 0
  assert torch.equal(out, w[:3].flip(-1, -2))
  u = torch.randint(0, 256, (2, 8, 8, 2), generator=g, device='cuda', dtype=torch.uint8)
  # a correctly rounded float32 division, as the reference's `inputs / 255` (the framework multiplies by the rounded reciprocal)
  assert torch.equal(qtrain.input_scale(u), (u.double() / 255.0).float())


@pytest.mark.parametrize('rf', [5, 4])
def test_hand_net_forward_and_backward_match_the_module_autograd(rf):
  """`HandNet` against `DeepQSiamFCN`'s own graph in float64 (Stack-v0 shapes and the 64 x 64 configuration): Q values and
  every parameter's gradient for a random upstream gradient, the backward restricted to the first samples of a larger
  saved forward (the update evaluates Q(s, .) and Q(s', .) in one pass)."""
  import copy
  from stackrl_amd import nets, qtrain
  h = 2 ** rf
  spec = ((4 * h, 4 * h, 2), (h, h, 1))
  net = nets.DeepQSiamFCN(spec, seed=3).cuda()
  gen = torch.Generator(device='cuda').manual_seed(rf)
  B, n = 5, 3
  xm = torch.randint(0, 256, (B, 4 * h, 4 * h, 2), generator=gen, device='cuda', dtype=torch.uint8)
  xm[..., 1] = (xm[..., 1] > 128).to(torch.uint8) * 170
  xo = torch.randint(0, 120, (B, h, h, 1), generator=gen, device='cuda', dtype=torch.uint8)
  ref = copy.deepcopy(net).double().cpu()                               # the module graph in float64 on the HOST
  fx, fx0 = ref.left(xm.cpu().permute(0, 3, 1, 2).double() / 255.0)    # models.py:144-147 in float64
  fw, _ = ref.right(xo.cpu().permute(0, 3, 1, 2).double() / 255.0)
  qd = ref.head(ref.correlation(fx, fw), fx0)
  gq = torch.randn((n, qd.shape[1]), generator=gen, device='cuda') / qd.shape[1] ** 0.5
  qd[:n].backward(gq.double().cpu())
  qd = qd.cuda()
  for p in net.parameters():
    p.grad = torch.full_like(p, float('nan'))      # the backward WRITES every element of every gradient (nothing accumulates,
  hn = qtrain.HandNet(net)                         # nothing needs a zero-fill first): a NaN left behind fails below
  hn.refresh()
  q = hn.forward((xm, xo), save=True)
  assert _rel(q.detach(), qd.detach()) <= 2e-4
  q0 = hn.forward((xm, xo))                                  # the no-grad evaluation is the same arithmetic
  assert torch.equal(q0, q.detach())
  hn.backward(gq)
  worst = 0.0
  for (name, p), pr in zip(net.named_parameters(), ref.parameters()):
    assert bool(torch.isfinite(p.grad).all()), name + ': a poisoned (never written) partial sum'
    if float(pr.grad.abs().max()) < 1e-12:          # the projection's bias cancels in A - mean(A): its gradient is zero
      assert float(p.grad.abs().max()) <= 1e-5, name          # float32 cancellation of 9,409 terms
      continue
    e = _rel(p.grad, pr.grad.cuda())
    worst = max(worst, e)
    assert e <= 2e-3, (name, e, _where(p.grad, pr.grad.cuda()))
  print('resolution factor', rf, 'worst relative parameter-gradient error', worst)
