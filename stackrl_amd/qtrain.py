"""Hand-written forward / backward of `DeepQSiamFCN` for the DQN minibatch update (`DQN.train`, stackrl/agents/dqn.py:397-476).

The reference differentiates the Q-network (stackrl/nets/models.py:106-201, `layers.unet` layers.py:135-259) with
TensorFlow's float32 convolutions.  Rounds 1-2 ran the update's convolutions through the library (MIOpen forward, data- and
weight-gradient kernels, ~940 launches per update).  `HandNet` runs every convolution of the update — forward with saved
activations, data gradient, weight gradient, for the 3 x 3 layers, the 2 x 2 transposed ones and the thin first layers — on
the kernels of csrc/train_conv.hip (true float32 on the matrix cores, fixed-order reductions), the cross-correlation and its
two gradients on csrc/xcorr_mfma.hip, the dueling head (1 x 1 projection + combination, `srl_thead_*`) and — since round 4 — its
value branch (average pool, two dense layers, `srl_tvalue_*`), the layout passes around the cross-correlation and the input
scaling on fixed-order kernels of the same file: no library or framework kernel runs inside forward() / backward().

Activations are float32 NHWC tensors ([B, H, W, C] contiguous); the decoder's concatenation buffers are written in place by
the producing kernels (channel slices), the max-pool gradient is routed inside the activation-gradient pass.  Weight
gradients land directly in `p.grad` (views of the DQN's flat gradient bucket — the all-reduce operand).  There is no CPU
fallback: the class needs the HIP extension.
"""
import ctypes

import torch

from stackrl_amd import qops

_F = torch.nn.functional


def _lib():
  L = qops.load()
  if not getattr(L, '_train_conv_ready', False):
    VP, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.srl_tconv.restype = ctypes.c_int
    L.srl_tconv.argtypes = [VP, I32, I32, VP, VP, VP, I32, I32] + [I32] * 8 + [VP]
    L.srl_twrw_scratch_floats.restype = I64
    L.srl_twrw_scratch_floats.argtypes = [I32] * 6
    L.srl_twrw.restype = ctypes.c_int
    L.srl_twrw.argtypes = [VP, I32, I32, VP, VP, VP] + [I32] * 7 + [VP, I32, I32, VP, VP]
    L.srl_tact_bwd_blocks.restype = I32
    L.srl_tact_bwd_blocks.argtypes = [I64, I32]
    L.srl_tact_bwd_scratch_floats.restype = I64
    L.srl_tact_bwd_scratch_floats.argtypes = [I64, I32]
    L.srl_tact_bwd.restype = ctypes.c_int
    L.srl_tact_bwd.argtypes = [VP, I32, I32, VP, I32, I32, VP, VP, VP, VP] + [I32] * 6 + [VP]
    L.srl_trepack.restype = ctypes.c_int
    L.srl_trepack.argtypes = [VP, VP, VP, I32, I64, VP]
    L.srl_thead_fwd.restype = ctypes.c_int
    L.srl_thead_fwd.argtypes = [VP, VP, VP, VP, VP, I32, I32, VP]
    L.srl_thead_bwd.restype = ctypes.c_int
    L.srl_thead_bwd.argtypes = [VP] * 8 + [I32, I32, VP]
    L.srl_tvalue_fwd.restype = ctypes.c_int
    L.srl_tvalue_fwd.argtypes = [VP] * 8 + [I32] * 4 + [VP]
    L.srl_tvalue_bwd.restype = ctypes.c_int
    L.srl_tvalue_bwd.argtypes = [VP] * 12 + [I32] * 4 + [VP]
    L.srl_tlayout.restype = ctypes.c_int
    L.srl_tlayout.argtypes = [VP, I32, I32, VP, I32, I32, I32, I32, VP]
    L.srl_tcorr_grad.restype = ctypes.c_int
    L.srl_tcorr_grad.argtypes = [VP, I32, VP, VP, I32, I32, I32, VP]
    L.srl_tflip.restype = ctypes.c_int
    L.srl_tflip.argtypes = [VP, VP, I64, I32, VP]
    L.srl_tu8_to_f32.restype = ctypes.c_int
    L.srl_tu8_to_f32.argtypes = [VP, VP, I64, VP]
    L.srl_train_conv_last_error.restype = ctypes.c_char_p
    L._train_conv_ready = True
  return L


def _chk(rc):
  if rc:
    raise RuntimeError(_lib().srl_train_conv_last_error().decode())


class Act(object):
  """A float32 NHWC activation that may be a channel slice of a wider buffer: pixel stride / channel offset in floats."""

  def __init__(self, t, C=None, off=0):
    self.t = t                                   # [B, H, W, stride] contiguous
    self.B, self.H, self.W, self.stride = (int(v) for v in t.shape)
    self.C = int(C) if C is not None else self.stride
    self.off = int(off)

  def ptr(self):
    return self.t.data_ptr()

  def first(self, n):
    """The first n samples (the batch is the outermost dimension)."""
    return Act(self.t[:n], self.C, self.off)

  def dense(self):
    """A contiguous [B, H, W, C] tensor of this activation (a view when it already is one)."""
    return self.t if (self.off == 0 and self.C == self.stride) else self.t[..., self.off:self.off + self.C].contiguous()


def tconv(x, wp, bias, cout, taps=9, relu=True, out=None, d2s=0):
  """`srl_tconv`: x `Act` -> `Act` of cout channels (a new tensor, or the channel slice `out` = (tensor, offset));
  d2s = cout_t: the transposed convolution's depth-to-space store into a map twice the size."""
  B, H, W = x.B, x.H, x.W
  co = d2s if d2s else cout
  if out is None:
    y = Act(torch.empty((B, 2 * H, 2 * W, co) if d2s else (B, H, W, co), dtype=torch.float32, device=x.t.device))
  else:
    y = Act(out[0], co, out[1])
  with torch.cuda.device(x.t.device):
    _chk(_lib().srl_tconv(x.ptr(), x.stride, x.off, wp.data_ptr(), None if bias is None else bias.data_ptr(), y.ptr(), y.stride,
                          y.off, B, H, W, x.C, cout, taps, int(bool(relu)), int(d2s), qops._stream(x.t)))
  return y


class _Scratch(object):
  """One growing float32 scratch buffer per (device, purpose): fixed addresses once the sizes have been seen.

  poison (a debug switch, off in the product; the GPU tests of the kernels turn it on): fill the buffer with NaN every time
  it is handed out, so that a partial-sum slot a finishing kernel reads without anybody having written it cannot pass as
  a plausible number."""
  poison = False

  def __init__(self):
    self.buf = {}

  def get(self, key, n, dev):
    t = self.buf.get(key)
    if t is None or t.numel() < n:
      t = torch.empty(int(n), dtype=torch.float32, device=dev)
      self.buf[key] = t
    if _Scratch.poison:
      t.fill_(float('nan'))
    return t


def twrw(x, gz, gw, scratch, taps=9, convt=False, bias=None):
  """`srl_twrw`: weight gradient of a layer with input `Act` x and activation gradient gz (contiguous [B, H, W, cout] tensor)
  into gw (the parameter's gradient tensor, framework layout).  bias = (partials, blocks, C, gbias) left by
  `tact_bwd(..., defer_bias=True)`: the same finishing launch writes the bias gradient."""
  B, H, W = x.B, x.H, x.W
  cout = int(gz.shape[-1])
  n = _lib().srl_twrw_scratch_floats(B, H, W, x.C, cout, taps)
  sc = scratch.get('wrw', n, gz.device)
  bp, bn, bc, gb = bias if bias is not None else (None, 0, 0, None)
  with torch.cuda.device(gz.device):
    _chk(_lib().srl_twrw(x.ptr(), x.stride, x.off, gz.data_ptr(), gw.data_ptr(), sc.data_ptr(), B, H, W, x.C, cout, taps,
                         int(bool(convt)), None if bp is None else bp.data_ptr(), int(bn), int(bc),
                         None if gb is None else gb.data_ptr(), qops._stream(gz)))


def tact_bwd(g, y, scratch, gbias=None, gpool=None, relu=True, s2d=False, defer_bias=False):
  """`srl_tact_bwd`: g, y `Act`s of the same shape -> gz tensor ([B, H, W, C], or space-to-depth [B, H/2, W/2, 4C]).
  defer_bias: leave the bias gradient as per-block partials and return (gz, (partials, blocks, C, gbias)) for `twrw`."""
  B, H, W, C = g.B, g.H, g.W, g.C
  dev = g.t.device
  gz = torch.empty((B, H // 2, W // 2, 4 * C) if s2d else (B, H, W, C), dtype=torch.float32, device=dev)
  sc = scratch.get('act', _lib().srl_tact_bwd_scratch_floats(B * H * W, C), dev) if gbias is not None else None
  with torch.cuda.device(dev):
    _chk(_lib().srl_tact_bwd(g.ptr(), g.stride, g.off, None if y is None else y.ptr(), 0 if y is None else y.stride,
                             0 if y is None else y.off, None if gpool is None else gpool.data_ptr(), gz.data_ptr(),
                             None if (gbias is None or defer_bias) else gbias.data_ptr(), None if sc is None else sc.data_ptr(),
                             B, H, W, C, int(bool(relu)), int(bool(s2d)), qops._stream(g.t)))
  if defer_bias:
    return gz, (sc, _lib().srl_tact_bwd_blocks(B * H * W, C), C, gbias)
  return gz


def to_nchw(a):
  """`srl_tlayout`: an `Act` -> contiguous channel-major tensor [B, C, H, W] (what the cross-correlation kernels read)."""
  out = torch.empty((a.B, a.C, a.H, a.W), dtype=torch.float32, device=a.t.device)
  with torch.cuda.device(a.t.device):
    _chk(_lib().srl_tlayout(a.ptr(), a.stride, a.off, out.data_ptr(), a.B, a.H * a.W, a.C, 0, qops._stream(a.t)))
  return out


def to_nhwc(t):
  """`srl_tlayout`: a contiguous channel-major tensor [B, C, H, W] -> `Act` [B, H, W, C]."""
  B, C, H, W = (int(v) for v in t.shape)
  out = torch.empty((B, H, W, C), dtype=torch.float32, device=t.device)
  with torch.cuda.device(t.device):
    _chk(_lib().srl_tlayout(t.data_ptr(), 0, 0, out.data_ptr(), B, H * W, C, 1, qops._stream(t)))
  return Act(out)


def input_scale(x):
  """The network's input scaling (models.py:144-147): uint8 observation [B, H, W, C] -> float32 / 255 (`srl_tu8_to_f32`)."""
  if x.dtype != torch.uint8 or x.numel() % 4:
    return (x.float() / 255.0) if x.dtype == torch.uint8 else x.float().contiguous()
  x = x.contiguous()
  out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
  with torch.cuda.device(x.device):
    _chk(_lib().srl_tu8_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), qops._stream(x)))
  return out


def pool2x2(y):
  """2 x 2 max-pool of an `Act` (csrc/epilogue.hip) -> contiguous [B, H/2, W/2, C]."""
  out = torch.empty((y.B, y.H // 2, y.W // 2, y.C), dtype=torch.float32, device=y.t.device)
  with torch.cuda.device(y.t.device):
    rc = qops.load().srl_pool2x2(y.ptr(), out.data_ptr(), y.B, y.H, y.W, y.C, y.stride, y.off, 1, qops._stream(y.t))
  if rc:
    raise RuntimeError(qops.load().srl_epilogue_last_error().decode())
  return Act(out)


class Packed(object):
  """The packed weight layouts of every convolution of a `DeepQSiamFCN` (srl_trepack), rebuilt from the parameters'
  flat bucket in ONE launch whenever the weights have changed."""

  def __init__(self, net, flat=None):
    self.net = net
    dev = next(net.parameters()).device
    self.convs = [m for m in net.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
    self.own_flat = flat is None
    self.flat = flat
    self._layout(dev)

  def _src(self, p):
    return (p.data_ptr() - self.flat.data_ptr()) // 4

  def _layout(self, dev):
    if self.own_flat:      # a net whose parameters are not views of one bucket (the target net): a private flat copy
      ws = [m.weight for m in self.convs]
      offs, o = [], 0
      for w in ws:
        offs.append(o); o += w.numel()
      self.flat = torch.empty(o, dtype=torch.float32, device=dev)
      self._own = list(zip(ws, offs))
    desc, self.view, o = [], {}, 0
    for i, m in enumerate(self.convs):
      w = m.weight
      src = self._own[i][1] if self.own_flat else self._src(w)
      if not self.own_flat:
        assert 0 <= src and src + w.numel() <= self.flat.numel(), 'parameter is not a view of the flat bucket'
      if isinstance(m, torch.nn.ConvTranspose2d):
        cin, cout = int(w.shape[0]), int(w.shape[1])
        for kind in (2, 3):
          n = cin * 4 * cout
          desc.append([src, o, cin, cout, 1, kind, 0, 0]); self.view[(m, kind)] = (o, n); o += n
      else:
        cout, cin, taps = int(w.shape[0]), int(w.shape[1]), int(w.shape[2] * w.shape[3])
        if cout % 16:
          continue                                # the 1 x 1 projection to one channel stays a framework op
        cin_p = (cin + 3) // 4 * 4
        n = taps * cin_p * cout
        desc.append([src, o, cin, cout, taps, 0, 0, 0]); self.view[(m, 0)] = (o, n); o += n
        cpad = (cin + 15) // 16 * 16             # data gradient: cin output channels, padded to a supported count
        n = taps * cout * cpad
        desc.append([src, o, cin, cout, taps, 1, cpad, 0]); self.view[(m, 1)] = (o, n); o += n
    self.total = o
    self.desc = torch.tensor(desc, dtype=torch.int64, device=dev)
    self.packed = torch.empty(o, dtype=torch.float32, device=dev)
    self.nlayers = len(desc)

  def refresh(self):
    if self.own_flat:
      with torch.no_grad():
        for w, off in self._own:
          self.flat[off:off + w.numel()].copy_(w.detach().reshape(-1))
    with torch.cuda.device(self.packed.device):
      _chk(_lib().srl_trepack(self.flat.data_ptr(), self.packed.data_ptr(), self.desc.data_ptr(), self.nlayers, self.total,
                              qops._stream(self.packed)))

  def w(self, m, kind):
    o, n = self.view[(m, kind)]
    return self.packed[o:o + n]


class HandNet(object):
  """Forward (optionally saving what the backward needs) and backward of a `DeepQSiamFCN` on the hand-written kernels.

  forward(inputs, save=False) -> Q float32 [B, A]                      (uint8 NHWC observations, as the env returns them)
  backward(grad_q)             -> parameter gradients into every `p.grad` (for the first `grad_q.shape[0]` samples of the
                                  saved forward)."""

  def __init__(self, net, flat=None, precision=qops.BF16X3):
    if not next(net.parameters()).is_cuda:
      raise RuntimeError('HandNet needs a HIP device: there is no CPU fallback')
    _lib()
    self.net = net
    self.packed = Packed(net, flat)
    self.scratch = _Scratch()
    self.precision = precision
    self.saved = None
    if len(net.pos) != 5 or not net.dueling or not net.dueling_avg_pool:
      raise ValueError('HandNet implements the config.gin network (pos_depth 2, dueling head with average pooling)')

  def refresh(self):
    """Re-pack the weights (call after every optimiser step / target sync)."""
    self.packed.refresh()
    self._versions = self._param_versions()

  def _param_versions(self):
    return tuple(p._version for p in self.net.parameters())

  def refresh_if_stale(self):
    """Re-pack when a parameter has been written since the last packing (in-place writes seen by the framework — a target
    sync, a restore; not the hipGraph replay of the optimiser step, after which the caller re-packs itself)."""
    if getattr(self, '_versions', None) != self._param_versions():
      self.refresh()

  # ---------------------------------------------------------------------------------------------- U-Net (layers.py:135-259)
  def _conv(self, m, x, out=None, relu=True):
    return tconv(x, self.packed.w(m, 0), m.bias, m.out_channels, taps=m.kernel_size[0] * m.kernel_size[1], relu=relu, out=out)

  def _unet_fwd(self, U, x, tape):
    cats = []
    for blk in U.down:
      f = blk[0].out_channels
      y1 = self._conv(blk[0], x)
      cat = torch.empty((x.B, x.H, x.W, 2 * f), dtype=torch.float32, device=x.t.device)
      y2 = self._conv(blk[2], y1, out=(cat, f))                       # skip -> second half of the decoder's concat buffer
      tape.append(('down', blk, x, y1, y2))
      cats.append(cat)
      x = pool2x2(y2)
    y1 = self._conv(U.bottom[0], x)
    x0 = self._conv(U.bottom[2], y1)
    tape.append(('bottom', U.bottom, x, y1, x0))
    x = x0
    for up, blk in zip(U.up, U.upconv):
      cat = cats.pop()
      f = up.out_channels
      yu = tconv(x, self.packed.w(up, 2), up.bias, 4 * f, taps=1, relu=True, out=(cat, 0), d2s=f)   # up{i} -> first half
      xc = Act(cat)
      y1 = self._conv(blk[0], xc)
      y2 = self._conv(blk[2], y1)
      tape.append(('up', (up, blk), x, yu, xc, y1, y2))
      x = y2
    return x, x0

  def _layer_bwd(self, m, x_in, y, g, n, gpool=None, need_dx=True, s2d=False):
    """One convolution backwards: g = gradient wrt the layer's output `y` (an `Act`), for the first n samples.  Writes the
    bias and weight gradients, returns the gradient wrt the input (an `Act`) or None."""
    gz, bias = tact_bwd(g, y.first(n), self.scratch, gbias=m.bias.grad, gpool=gpool, relu=True, s2d=s2d, defer_bias=True)
    if isinstance(m, torch.nn.ConvTranspose2d):
      xin = x_in.first(n)
      twrw(xin, gz, m.weight.grad, self.scratch, taps=1, convt=True, bias=bias)
      if not need_dx:
        return None
      return tconv(Act(gz), self.packed.w(m, 3), None, m.in_channels, taps=1, relu=False)
    taps = m.kernel_size[0] * m.kernel_size[1]
    twrw(x_in.first(n), gz, m.weight.grad, self.scratch, taps=taps, bias=bias)
    if not need_dx:
      return None
    cpad = (m.in_channels + 15) // 16 * 16
    gx = tconv(Act(gz), self.packed.w(m, 1), None, cpad, taps=taps, relu=False)
    return Act(gx.t, m.in_channels, 0) if cpad != m.in_channels else gx

  def _unet_bwd(self, U, tape, g_out, g_x0, n, need_dx_input=False):
    """g_out: gradient wrt the U-Net's output (`Act`); g_x0: None, or a callable that takes the gradient arriving at the bottom
    features through the decoder and returns the full gradient there (the value branch of the dueling head joins in)."""
    ups = [t for t in tape if t[0] == 'up']
    downs = [t for t in tape if t[0] == 'down']
    bottom = [t for t in tape if t[0] == 'bottom'][0]
    g = g_out
    gcats = []
    for (_, (up, blk), x_prev, yu, xc, y1, y2) in reversed(ups):
      g = self._layer_bwd(blk[2], y1, y2, g, n)
      gcat = self._layer_bwd(blk[0], xc, y1, g, n)                   # [n, H, W, 2f]: first half -> up{i}, second half -> the skip
      f = up.out_channels
      gcats.append(gcat)
      g = self._layer_bwd(up, x_prev, yu, Act(gcat.t, f, 0), n, s2d=True)
    if g_x0 is not None:
      g = g_x0(g)
    (_, bm, xb, y1b, x0) = bottom
    g = self._layer_bwd(bm[2], y1b, x0, g, n)
    g = self._layer_bwd(bm[0], xb, y1b, g, n)                        # gradient wrt the deepest pooled map
    for k, (_, blk, x_in, y1, y2) in enumerate(reversed(downs)):
      gcat = gcats[len(downs) - 1 - k] if gcats else None
      f = blk[0].out_channels
      # the skip's gradient: its half of the concat gradient + what comes back through the 2 x 2 max-pool
      gsk = Act(gcat.t, f, f)
      g = self._layer_bwd(blk[2], y1, y2, gsk, n, gpool=g.dense())
      last = k == len(downs) - 1
      g = self._layer_bwd(blk[0], x_in, y1, g, n, need_dx=(not last) or need_dx_input)
    return g

  # ---------------------------------------------------------------------------------------------- the whole net
  def forward(self, inputs, save=False):
    net = self.net
    xm, xo = inputs
    B = int(xm.shape[0])
    tape_l, tape_r = [], []
    xl, xr = Act(input_scale(xm)), Act(input_scale(xo))                                          # models.py:144-147
    fl, x0 = self._unet_fwd(net.left, xl, tape_l)
    fr, _ = self._unet_fwd(net.right, xr, tape_r)
    # layers.correlation (layers.py:21-38) on the matrix cores; the kernel reads channel-major tensors
    xl_n, xr_n = to_nchw(fl), to_nchw(fr)
    corr = qops.xcorr_forward_mfma(xl_n, xr_n, self.precision)                                   # [B, 1, O, O]
    O = int(corr.shape[-1])
    cin = Act(corr.reshape(B, O, O, 1))
    z1 = self._conv(net.pos[0], cin)
    z2 = self._conv(net.pos[2], z1)
    # the value branch (layers.value: average pool of the bottom features -> Dense + ReLU -> Dense) and the 1 x 1 projection
    # with the dueling combination (models.py:179-192): hand-written passes, one workgroup per sample (srl_tvalue_fwd,
    # srl_thead_fwd)
    d1, d2 = net.value[0], net.value[2]
    C, U, P = int(d1.in_features), int(d1.out_features), x0.H * x0.W
    x0d = x0.dense()
    pooled = torch.empty((B, C), dtype=torch.float32, device=xm.device) if save else None
    hid = torch.empty((B, U), dtype=torch.float32, device=xm.device) if save else None
    v = torch.empty((B,), dtype=torch.float32, device=xm.device)
    pw, pb = net.pos[4].weight.detach().reshape(-1), net.pos[4].bias.detach()
    q = torch.empty((B, O * O), dtype=torch.float32, device=xm.device)
    with torch.cuda.device(q.device):
      _chk(_lib().srl_tvalue_fwd(x0d.data_ptr(), d1.weight.data_ptr(), d1.bias.data_ptr(), d2.weight.data_ptr(), d2.bias.data_ptr(),
                                 None if pooled is None else pooled.data_ptr(), None if hid is None else hid.data_ptr(),
                                 v.data_ptr(), B, P, C, U, qops._stream(q)))
      _chk(_lib().srl_thead_fwd(z2.t.data_ptr(), pw.data_ptr(), pb.data_ptr(), v.data_ptr(), q.data_ptr(), B, O * O, qops._stream(q)))
    if save:
      self.saved = dict(tape_l=tape_l, tape_r=tape_r, fl=fl, fr=fr, xl_n=xl_n, xr_n=xr_n, cin=cin, z1=z1, z2=z2,
                        pooled=pooled, hid=hid, x0=x0, B=B, O=O)
    return q

  def backward(self, grad_q):
    """Gradients of sum(q[:n] * grad_q) wrt every parameter, written into `p.grad` (n = grad_q.shape[0] <= the saved batch)."""
    S, net = self.saved, self.net
    n = int(grad_q.shape[0])
    O = S['O']
    A = O * O
    dev = grad_q.device
    z2 = S['z2'].t                                                      # [B, O, O, 16], contiguous
    pj, d1, d2 = net.pos[4], net.value[0], net.value[2]
    for p_ in (pj.weight, pj.bias, d1.weight, d1.bias, d2.weight, d2.bias):
      if p_.grad is None:
        p_.grad = torch.zeros_like(p_)
    gz2 = torch.empty((n, O, O, 16), dtype=torch.float32, device=dev)
    gv = torch.empty((n,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
      _chk(_lib().srl_thead_bwd(z2.data_ptr(), pj.weight.detach().reshape(-1).data_ptr(), grad_q.contiguous().data_ptr(),
                                gz2.data_ptr(), gv.data_ptr(), pj.weight.grad.data_ptr(), pj.bias.grad.data_ptr(),
                                self.scratch.get('head', 17 * n, dev).data_ptr(), n, A, qops._stream(gz2)))
    g = self._layer_bwd(net.pos[2], S['z1'], S['z2'], Act(gz2), n)
    g = self._layer_bwd(net.pos[0], S['cin'], S['z1'], g, n)                     # [n, O, O, 16], channel 0 = d / d corr
    kh = int(S['xr_n'].shape[-1])
    H = int(S['xl_n'].shape[-1])
    C = int(S['xl_n'].shape[1])
    # d / d corr as the plain map and zero-padded (the operand of the data gradient), the flipped kernels: one pass each
    gcorr = torch.empty((n, O, O), dtype=torch.float32, device=dev)
    gp = torch.empty((n, O + 2 * (kh - 1), O + 2 * (kh - 1)), dtype=torch.float32, device=dev)
    wflip = torch.empty((n, C, kh, kh), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
      _chk(_lib().srl_tcorr_grad(g.t.data_ptr(), g.stride, gcorr.data_ptr(), gp.data_ptr(), n, O, kh - 1, qops._stream(gp)))
      _chk(_lib().srl_tflip(S['xr_n'].data_ptr(), wflip.data_ptr(), n * C, kh * kh, qops._stream(gp)))
    dxl = qops._xcorr_mfma(1, self.precision, gp, wflip, n, C, H, kh)                           # [n, C, H, H]
    dxr = qops._xcorr_mfma(2, self.precision, S['xl_n'][:n], gcorr, n, C, H, kh)              # [n, C, kh, kh]
    g_l, g_r = to_nhwc(dxl), to_nhwc(dxr)

    def value_branch(g_dec):
      """The gradient wrt the bottom features: what arrives through the decoder + the value branch's (srl_tvalue_bwd, which
      also writes the two dense layers' gradients)."""
      x0 = S['x0']
      Cb, U, P = int(d1.in_features), int(d1.out_features), x0.H * x0.W
      gin = g_dec.dense()
      gx = torch.empty((n, x0.H, x0.W, Cb), dtype=torch.float32, device=dev)
      with torch.cuda.device(dev):
        _chk(_lib().srl_tvalue_bwd(gv.data_ptr(), S['hid'].data_ptr(), S['pooled'].data_ptr(), d1.weight.data_ptr(),
                                   d2.weight.data_ptr(), gin.data_ptr(), gx.data_ptr(), d1.weight.grad.data_ptr(),
                                   d1.bias.grad.data_ptr(), d2.weight.grad.data_ptr(), d2.bias.grad.data_ptr(),
                                   self.scratch.get('value', n * U, dev).data_ptr(), n, P, Cb, U, qops._stream(gx)))
      return Act(gx)

    self._unet_bwd(net.left, S['tape_l'], g_l, value_branch, n)
    self._unet_bwd(net.right, S['tape_r'], g_r, None, n)
    self.saved = None
