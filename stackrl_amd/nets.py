"""Q-network of the hot path: `DeepQSiamFCN` (stackrl/nets/models.py:106-201) with the layer blocks of
stackrl/nets/layers.py — `unet` (:135-259), `correlation` (:21-38), `pos_layers` (:439-472), `value` (:424-436).

Pseudo-siamese fully convolutional net: a "left" U-Net on the (H, W, 2) height+goal map, a "right" U-Net on the
(h, w, 1) object map, per-sample VALID cross-correlation of the two feature maps summed over channels
(-> one value per placement pixel), two 3x3 convs + a 1x1, flatten -> advantages; dueling value head on the
left bottleneck; Q = A - mean(A) + V (models.py:179-192).

Plain convolutions go through MIOpen (PyTorch-ROCm, MFMA for bf16/fp16).  The cross-correlation — the one op of
this net that is not a library conv — has a hand-written HIP forward for the rollout path (stackrl_amd/qops.py,
csrc/qnet.hip); training differentiates the same op through a grouped convolution.

Inputs are the env's uint8 NHWC tensors (scaled by 1/255, models.py:144-147); outputs are float32 [B, A].
"""
import math
import random

import torch
from torch import nn
import torch.nn.functional as F


def he_normal_(tensor, fan_in, generator=None):
  """Keras `he_normal` (layers.py:9-18): truncated normal at 2 sigma, variance 2/fan_in after truncation."""
  std = math.sqrt(2.0 / fan_in) / .87962566103423978
  with torch.no_grad():
    return nn.init.trunc_normal_(tensor, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=generator)


def _init_conv(m, gen):
  if isinstance(m, (nn.Conv2d, nn.Linear)):
    w = m.weight
    fan_in = w[0].numel()                       # in_channels * kh * kw (Keras fan_in)
    he_normal_(w, fan_in, gen)
    nn.init.zeros_(m.bias)
  elif isinstance(m, nn.ConvTranspose2d):
    # torch [in, out, kh, kw]; the Keras Conv2DTranspose kernel is (kh, kw, out, in) and `_compute_fans` takes
    # fan_in = shape[-2] * receptive field = out_channels * kh * kw (layers.py:9-18 he_normal on that kernel)
    w = m.weight
    fan_in = w.shape[1] * w.shape[2] * w.shape[3]
    he_normal_(w, fan_in, gen)
    nn.init.zeros_(m.bias)


def _fused_conv(m, x, relu=True):
  """Convolution without bias + the fused bias / ReLU pass (and its hand-written backward) of csrc/epilogue.hip."""
  from stackrl_amd import qops
  if not qops.bias_act_supported(m.out_channels):
    # channel counts the fused pass is not built for (above 256, or C / 8 not dividing 256 — e.g. a 512-channel bottom
    # with left_filters=32): the module path, as before the fused passes existed
    y = m(x)
    return F.relu(y) if relu else y
  if isinstance(m, nn.ConvTranspose2d):
    y = F.conv_transpose2d(x, m.weight, None, stride=m.stride)
  else:
    y = F.conv2d(x, m.weight, None, padding=m.padding)
  return qops.bias_act_autograd(y, m.bias, relu)


def _use_fused(module, x):
  return getattr(module, 'fused_epilogues', False) and x.is_cuda and x.dtype == torch.float32 and \
    not torch.is_autocast_enabled()


class UNet(nn.Module):
  """`layers.unet` with `double_endpoint=True`, `out_channels=None` (layers.py:135-259)."""

  def __init__(self, in_channels, depth, filters):
    super().__init__()
    self.depth = depth
    self.down = nn.ModuleList()
    c = in_channels
    for i in range(depth):                      # convdw{i}{0,1}
      f = filters * 2 ** i
      self.down.append(nn.Sequential(nn.Conv2d(c, f, 3, padding=1), nn.ReLU(inplace=True),
                                     nn.Conv2d(f, f, 3, padding=1), nn.ReLU(inplace=True)))
      c = f
    f = filters * 2 ** depth                    # conv{depth}{0,1}
    self.bottom = nn.Sequential(nn.Conv2d(c, f, 3, padding=1), nn.ReLU(inplace=True),
                                nn.Conv2d(f, f, 3, padding=1), nn.ReLU(inplace=True))
    c = f
    self.up = nn.ModuleList()
    self.upconv = nn.ModuleList()
    for i in range(depth - 1, -1, -1):          # up{i}, concat{i}, convuw{i}{0,1}
      f = filters * 2 ** i
      self.up.append(nn.ConvTranspose2d(c, f, 2, stride=2))
      self.upconv.append(nn.Sequential(nn.Conv2d(2 * f, f, 3, padding=1), nn.ReLU(inplace=True),
                                       nn.Conv2d(f, f, 3, padding=1), nn.ReLU(inplace=True)))
      c = f
    self.out_channels = c
    self.bottom_channels = filters * 2 ** depth

  def forward(self, x):
    if _use_fused(self, x):
      return self._forward_fused(x)
    levels = []
    for blk in self.down:
      x = blk(x)
      levels.append(x)
      x = F.max_pool2d(x, 2)
    x = self.bottom(x)
    x0 = x
    for up, blk in zip(self.up, self.upconv):
      x = F.relu(up(x))
      x = torch.cat([x, levels.pop()], dim=1)   # Concatenate([x, skip]), layers.py:231
      x = blk(x)
    return x, x0


  def _forward_fused(self, x):
    """The same graph with every bias add + ReLU (and their backward + the bias-gradient reduction) as one hand-written
    pass each, on float32 channels-last tensors (the GPU update path of `DQN.train`)."""
    levels = []
    for blk in self.down:
      x = _fused_conv(blk[2], _fused_conv(blk[0], x))
      levels.append(x)
      x = F.max_pool2d(x, 2)
    x = _fused_conv(self.bottom[2], _fused_conv(self.bottom[0], x))
    x0 = x
    for up, blk in zip(self.up, self.upconv):
      x = _fused_conv(up, x)
      x = torch.cat([x, levels.pop()], dim=1)
      x = _fused_conv(blk[2], _fused_conv(blk[0], x))
    return x, x0

  fused_epilogues = False


def correlation_reference(x, w):
  """`layers.correlation` (layers.py:21-38): per-sample VALID cross-correlation summed over channels.
  x [B, C, H, W], w [B, C, h, w] -> [B, 1, H-h+1, W-w+1].  Differentiable (grouped convolution)."""
  B, C, H, W = x.shape
  out = F.conv2d(x.reshape(1, B * C, H, W), w, groups=B)
  return out.reshape(B, 1, out.shape[-2], out.shape[-1])


class DeepQSiamFCN(nn.Module):
  """models.py:106-201 with the `config.gin:55-59` values as defaults."""

  def __init__(self, input_spec=((128, 128, 2), (32, 32, 1)), left_filters=16, left_depth=4,
               right_filters=None, right_depth=None, pos_filters=16, pos_depth=2, dueling=True,
               dueling_avg_pool=True, dueling_units=256, seed=None):
    super().__init__()
    (H, W, cl), (h, w, cr) = [tuple(getattr(s, 'shape', s)) for s in input_spec]
    right_filters = right_filters or left_filters           # models.py:155
    right_depth = right_depth or max(1, left_depth - 2)     # models.py:156
    if right_filters != left_filters:
      raise ValueError('corr_channels (1x1 projection) is not implemented: use equal filters (config.gin default).')
    self.left = UNet(cl, left_depth, left_filters)
    self.right = UNet(cr, right_depth, right_filters)
    self.dueling = dueling
    self.dueling_avg_pool = dueling_avg_pool
    if dueling:                                             # layers.value, layers.py:424-436
      self.value = nn.Sequential(nn.Linear(self.left.bottom_channels, dueling_units), nn.ReLU(inplace=True),
                                 nn.Linear(dueling_units, 1))
    pos = []
    c = 1
    for _ in range(pos_depth):                              # layers.pos_layers, layers.py:439-472
      pos += [nn.Conv2d(c, pos_filters, 3, padding=1), nn.ReLU(inplace=True)]
      c = pos_filters
    pos.append(nn.Conv2d(c, 1, 1))
    self.pos = nn.Sequential(*pos)
    self.in_hw = ((H, W), (h, w))
    self.out_hw = (H - h + 1, W - w + 1)
    self.n_actions = self.out_hw[0] * self.out_hw[1]
    self.correlation = correlation_reference                # swapped for the HIP op on the rollout path
    # seeds chained like models.py:149-153 / layers.py:9-18 (python Random -> per-layer generators)
    r = random.Random(seed) if seed is not None else random.Random()
    for blk in (self.left, self.right, getattr(self, 'value', None), self.pos):
      if blk is None:
        continue
      rr = random.Random(r.randint(0, 2 ** 32 - 1))
      for m in blk.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
          gen = torch.Generator().manual_seed(rr.randint(0, 2 ** 32 - 1))
          _init_conv(m, gen)

  @staticmethod
  def prepare(inputs):
    """uint8 NHWC -> float NCHW scaled by 1/255 (`i/i.dtype.max`, models.py:144-147)."""
    x, w = inputs
    x = x.permute(0, 3, 1, 2).float() / 255.0 if x.dtype == torch.uint8 else x.permute(0, 3, 1, 2).float()
    w = w.permute(0, 3, 1, 2).float() / 255.0 if w.dtype == torch.uint8 else w.permute(0, 3, 1, 2).float()
    return x, w

  def features(self, inputs):
    x, w = self.prepare(inputs)
    x, x0 = self.left(x)
    w, _ = self.right(w)
    return x, x0, w

  def set_fused_epilogues(self, flag=True):
    """GPU update path: bias add + ReLU of every convolution as one hand-written pass with its own backward
    (csrc/epilogue.hip) instead of the library's separate kernels.  Needs the HIP extension."""
    if flag:
      from stackrl_amd import qops
      qops.load()
    self.fused_epilogues = self.left.fused_epilogues = self.right.fused_epilogues = bool(flag)
    return self

  fused_epilogues = False

  def _pos(self, corr):
    if _use_fused(self, corr) and len(self.pos) >= 3:
      x = corr
      for k in range(0, len(self.pos) - 1, 2):              # (conv, relu) pairs; the final 1 x 1 projection stays a module
        x = _fused_conv(self.pos[k], x)
      return self.pos[-1](x)
    return self.pos(corr)

  def head(self, corr, x0):
    a = self._pos(corr).flatten(1)                          # Flatten -> advantages
    if not self.dueling:
      return a
    pooled = x0.mean(dim=(2, 3)) if self.dueling_avg_pool else x0.amax(dim=(2, 3))
    v = self.value(pooled)
    return a - a.mean(dim=-1, keepdim=True) + v             # models.py:188-192

  def forward(self, inputs):
    x, x0, w = self.features(inputs)
    return self.head(self.correlation(x, w), x0)


def count_parameters(net):
  return sum(p.numel() for p in net.parameters())


def forward_macs(H=128, h=32, lf=16, ld=4, rd=2, pf=16):
  """Multiply-accumulates of one forward pass per sample (derived from the layer shapes)."""
  def unet(res, cin, depth, f):
    m, c, r = 0, cin, res
    for i in range(depth):
      fi = f * 2 ** i
      m += r * r * 9 * (c * fi + fi * fi); c = fi; r //= 2
    fb = f * 2 ** depth
    m += r * r * 9 * (c * fb + fb * fb); c = fb
    for i in range(depth - 1, -1, -1):
      fi = f * 2 ** i
      r *= 2
      m += r * r * c * fi                      # 2x2 stride-2 transposed conv: one tap per output pixel
      m += r * r * 9 * (2 * fi * fi + fi * fi); c = fi
    return m
  o = H - h + 1
  return dict(left=unet(H, 2, ld, lf), right=unet(h, 1, rd, lf), xcorr=o * o * h * h * lf,
              pos=o * o * (9 * pf + 9 * pf * pf + pf))
