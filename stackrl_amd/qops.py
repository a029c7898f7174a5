"""Python side of libstackrl_qnet.so (include/stackrl_qnet.h): the hand-written ops of the Q-net rollout path.
No CPU fallback: these functions need a HIP device and the built library."""
import ctypes
import os

import torch

from stackrl_amd import build as _build

_LIB = None


def load():
  global _LIB
  if _LIB is None:
    if not os.path.isfile(_build.QLIB):
      _build.build()
    L = ctypes.CDLL(_build.QLIB)
    VP = ctypes.c_void_p
    L.srl_xcorr_forward.restype = ctypes.c_int
    L.srl_xcorr_forward.argtypes = [VP, VP, VP] + [ctypes.c_int32] * 6 + [VP]
    L.srl_xcorr_mfma_scratch_bytes.restype = ctypes.c_int64
    L.srl_xcorr_mfma_scratch_bytes.argtypes = [ctypes.c_int32] * 6
    L.srl_xcorr_mfma.restype = ctypes.c_int
    L.srl_xcorr_mfma.argtypes = [ctypes.c_int32, ctypes.c_int32, VP, ctypes.c_int32, VP, ctypes.c_int32, VP, VP,
                                 ctypes.c_int64] + [ctypes.c_int32] * 4 + [VP]
    L.srl_xcorr_mfma_last_error.restype = ctypes.c_char_p
    L.srl_bias_act.restype = ctypes.c_int
    L.srl_bias_act.argtypes = [VP, VP, VP, ctypes.c_int64] + [ctypes.c_int32] * 5 + [VP]
    L.srl_bias_act_pool.restype = ctypes.c_int
    L.srl_bias_act_pool.argtypes = [VP, VP, VP, VP] + [ctypes.c_int32] * 6 + [VP]
    L.srl_bias_act_f32.restype = ctypes.c_int
    L.srl_bias_act_f32.argtypes = [VP, VP, VP, ctypes.c_int64] + [ctypes.c_int32] * 5 + [VP]
    L.srl_bias_act_pool_f32.restype = ctypes.c_int
    L.srl_bias_act_pool_f32.argtypes = [VP, VP, VP, VP] + [ctypes.c_int32] * 6 + [VP]
    L.srl_epilogue_last_error.restype = ctypes.c_char_p
    L.srl_conv3x3_wfrag_elems.restype = ctypes.c_int32
    L.srl_conv3x3_wfrag_elems.argtypes = [ctypes.c_int32] * 2
    L.srl_conv3x3_bias_relu.restype = ctypes.c_int
    L.srl_conv3x3_bias_relu.argtypes = [VP] * 5 + [ctypes.c_int32] * 8 + [VP]
    L.srl_conv3x3_bias_relu_f32.restype = ctypes.c_int
    L.srl_conv3x3_bias_relu_f32.argtypes = [VP] * 5 + [ctypes.c_int32] * 8 + [VP]
    L.srl_conv_last_error.restype = ctypes.c_char_p
    L.srl_convt2x2_wfrag_elems.restype = ctypes.c_int32
    L.srl_convt2x2_wfrag_elems.argtypes = [ctypes.c_int32] * 2
    L.srl_convt2x2_bias_relu.restype = ctypes.c_int
    L.srl_convt2x2_bias_relu.argtypes = [VP] * 4 + [ctypes.c_int32] * 7 + [VP]
    L.srl_conv3x3_thin.restype = ctypes.c_int
    L.srl_conv3x3_thin.argtypes = [VP, ctypes.c_int32, VP, VP, VP] + [ctypes.c_int32] * 6 + [VP]
    L.srl_bias_act_bwd_scratch_floats.restype = ctypes.c_int64
    L.srl_bias_act_bwd_scratch_floats.argtypes = [ctypes.c_int64, ctypes.c_int32]
    L.srl_bias_act_bwd_f32.restype = ctypes.c_int
    L.srl_bias_act_bwd_f32.argtypes = [VP] * 5 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, VP]
    L.srl_pool2x2.restype = ctypes.c_int
    L.srl_pool2x2.argtypes = [VP, VP] + [ctypes.c_int32] * 7 + [VP]
    L.srl_convt2x2_gemm_supported.restype = ctypes.c_int32
    L.srl_convt2x2_gemm_supported.argtypes = [ctypes.c_int32] * 2
    L.srl_convt2x2_gemm_bias_relu.restype = ctypes.c_int
    L.srl_convt2x2_gemm_bias_relu.argtypes = [VP] * 4 + [ctypes.c_int32] * 8 + [VP]
    L.srl_conv3x3_gemm_supported.restype = ctypes.c_int32
    L.srl_conv3x3_gemm_supported.argtypes = [ctypes.c_int32] * 3
    L.srl_conv3x3_gemm_batch_multiple.restype = ctypes.c_int32
    L.srl_conv3x3_gemm_batch_multiple.argtypes = [ctypes.c_int32] * 2
    L.srl_conv3x3_gemm_wfrag_elems.restype = ctypes.c_int64
    L.srl_conv3x3_gemm_wfrag_elems.argtypes = [ctypes.c_int32] * 2
    L.srl_conv3x3_gemm_bias_relu.restype = ctypes.c_int
    L.srl_conv3x3_gemm_bias_relu.argtypes = [VP] * 4 + [ctypes.c_int32] * 7 + [VP]
    L.srl_conv_gemm_last_error.restype = ctypes.c_char_p
    L.srl_conv3x3_thin_f32.restype = ctypes.c_int
    L.srl_conv3x3_thin_f32.argtypes = L.srl_conv3x3_thin.argtypes
    L.srl_convt2x2_bias_relu_f32.restype = ctypes.c_int
    L.srl_convt2x2_bias_relu_f32.argtypes = L.srl_convt2x2_bias_relu.argtypes
    L.srl_conv3x3_relu_project_f32.restype = ctypes.c_int
    L.srl_conv3x3_relu_project_f32.argtypes = [VP, VP, VP, VP, ctypes.c_float, VP] + [ctypes.c_int32] * 5 + [VP]
    L.srl_thin_conv3x3_bias_relu_f32.restype = ctypes.c_int
    L.srl_thin_conv3x3_bias_relu_f32.argtypes = [VP, ctypes.c_int32, ctypes.c_int32] + [VP] * 6 + [ctypes.c_int32] * 6 + [VP]
    L.srl_thin_conv3x3_relu_project_f32.restype = ctypes.c_int
    L.srl_thin_conv3x3_relu_project_f32.argtypes = [VP] * 6 + [ctypes.c_float, VP] + [ctypes.c_int32] * 3 + [VP]
    L.srl_conv3x3_relu_project.restype = ctypes.c_int
    L.srl_conv3x3_relu_project.argtypes = [VP, VP, VP, VP, ctypes.c_float, VP] + [ctypes.c_int32] * 5 + [VP]
    L.srl_policy_head.restype = ctypes.c_int
    L.srl_policy_head.argtypes = [VP, VP, VP, ctypes.c_float, VP, ctypes.c_int32, ctypes.c_int32, VP]
    L.srl_qnet_last_error.restype = ctypes.c_char_p
    I32, I64, F = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    L.srl_td_epilogue.restype = ctypes.c_int
    L.srl_td_epilogue.argtypes = [VP] * 7 + [F, F, F, I32, F, I32, I32] + [VP] * 7 + [VP]
    L.srl_adam_step.restype = ctypes.c_int
    L.srl_adam_step.argtypes = [VP, VP, VP, VP, I64, VP, F, F, F, F, VP]
    L.srl_gumbel_topk_scratch_bytes.restype = I64
    L.srl_gumbel_topk_scratch_bytes.argtypes = [I64, I32]
    L.srl_gumbel_topk.restype = ctypes.c_int
    L.srl_gumbel_topk.argtypes = [VP, VP, VP, I64, I32, VP, VP, VP, I64, VP]
    L.srl_replay_scatter.restype = ctypes.c_int
    L.srl_replay_scatter.argtypes = [VP, VP, I64, I64, VP, VP, VP, I32, I64, I64] + [VP] * 6 + [VP]
    L.srl_replay_gather.restype = ctypes.c_int
    L.srl_replay_gather.argtypes = [VP, I32, I64, I64, I32, VP, VP, VP, I64, I64] + [VP] * 15 + [VP]
    L.srl_logit_extrema_scratch_bytes.restype = I64
    L.srl_logit_extrema_scratch_bytes.argtypes = []
    L.srl_logit_extrema.restype = ctypes.c_int
    L.srl_logit_extrema.argtypes = [VP, I64, VP, VP, VP, VP]
    L.srl_learner_last_error.restype = ctypes.c_char_p
    _LIB = L
  return _LIB


def _stream(t):
  return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


_SCRATCH = {}
MFMA_SHAPES = ((128, 32), (64, 16))   # forward (H, kh) shapes built in csrc/xcorr_mfma.hip
BF16, BF16X3 = 0, 1                   # precisions of the MFMA path (include/stackrl_qnet.h)


def _xcorr_mfma(mode, precision, a, k, B, C, H, kh):
  """One launch of srl_xcorr_mfma (see the header for the three modes).  a / k: contiguous bf16 or fp32 tensors."""
  if not a.is_cuda:
    raise RuntimeError('the MFMA cross-correlation needs a HIP device (no CPU fallback)')
  if precision == BF16X3:
    a = a.float(); k = k.float()
  a = a.contiguous(); k = k.contiguous()
  for t in (a, k):
    if t.dtype not in (torch.float32, torch.bfloat16):
      raise TypeError('xcorr operands must be float32 or bfloat16')
  L = load()
  need = L.srl_xcorr_mfma_scratch_bytes(mode, precision, B, C, H, kh)
  if need < 0:
    raise RuntimeError('MFMA cross-correlation: unsupported shape (H=%d, kh=%d)' % (H, kh))
  key = (a.device.index, torch.cuda.current_stream(a.device).cuda_stream)
  scratch = _SCRATCH.get(key)
  if scratch is None or scratch.numel() < need:
    scratch = torch.empty(need, dtype=torch.uint8, device=a.device)
    _SCRATCH[key] = scratch
  O = H - kh + 1
  shape = {0: (B, 1, O, O), 1: (B, C, H, H), 2: (B, C, kh, kh)}[mode]
  out = torch.empty(shape, dtype=torch.float32, device=a.device)
  with torch.cuda.device(a.device):
    rc = L.srl_xcorr_mfma(mode, precision, a.data_ptr(), int(a.dtype == torch.float32), k.data_ptr(),
                          int(k.dtype == torch.float32), out.data_ptr(), scratch.data_ptr(), scratch.numel(),
                          B, C, H, kh, _stream(a))
  if rc:
    raise RuntimeError(L.srl_xcorr_mfma_last_error().decode())
  return out


def _mfma_ok(x, w):
  return x.dim() == 4 and x.shape[-1] == x.shape[-2] and w.shape[-1] == w.shape[-2] and \
         (x.shape[-1], w.shape[-1]) in MFMA_SHAPES


def xcorr_forward_mfma(x, w, precision=BF16):
  """`layers.correlation` forward on the matrix cores: x [B,C,H,H], w [B,C,kh,kh] -> float32 [B,1,OH,OW]."""
  B, C, H, _ = x.shape
  return _xcorr_mfma(0, precision, x, w, B, C, H, w.shape[-1])


class _XCorrMFMA(torch.autograd.Function):
  """Differentiable `layers.correlation` (layers.py:21-38) on the matrix cores: forward and both gradients are the
  same Toeplitz-MFMA kernel family (csrc/xcorr_mfma.hip)."""

  @staticmethod
  def forward(ctx, x, w, precision):
    ctx.save_for_backward(x, w)
    ctx.precision = precision
    return xcorr_forward_mfma(x, w, precision)

  @staticmethod
  def backward(ctx, g):
    x, w = ctx.saved_tensors
    B, C, H, _ = x.shape
    kh = w.shape[-1]
    g = g[:, 0].float()
    dx = dw = None
    if ctx.needs_input_grad[0]:
      gp = torch.nn.functional.pad(g, (kh - 1, kh - 1, kh - 1, kh - 1))
      dx = _xcorr_mfma(1, ctx.precision, gp, w.flip(-1, -2), B, C, H, kh).to(x.dtype)
    if ctx.needs_input_grad[1]:
      dw = _xcorr_mfma(2, ctx.precision, x, g, B, C, H, kh).to(w.dtype)
    return dx, dw, None


def correlation(precision=BF16X3):
  """A drop-in for `nets.correlation_reference` (assign to `net.correlation`): MFMA kernels for the shapes in
  MFMA_SHAPES on a HIP device, the library formulation otherwise."""
  from stackrl_amd import nets

  def f(x, w):
    if x.is_cuda and _mfma_ok(x, w):
      return _XCorrMFMA.apply(x, w, precision)
    return nets.correlation_reference(x, w)
  return f


def xcorr_forward(x, w, precision=None):
  """`layers.correlation` forward (layers.py:21-38): x [B,C,H,W], w [B,C,kh,kw] -> float32 [B,1,OH,OW].
  For the shapes in MFMA_SHAPES the matrix-core kernel runs: bfloat16 features (the autocast rollout path) with
  operands as they are, float32 features with the hi/lo split (bf16x3: fp32-class accuracy, same stated tolerance as
  the vector kernel).  Other shapes, or precision='fp32', take the fp32 vector kernel."""
  if not x.is_cuda:
    raise RuntimeError('xcorr_forward needs a HIP device (no CPU fallback)')
  if precision != 'fp32' and _mfma_ok(x, w):
    if x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16:
      return xcorr_forward_mfma(x, w, BF16)
    if x.dtype == torch.float32 and w.dtype == torch.float32:
      return xcorr_forward_mfma(x, w, BF16X3)
  x = x.contiguous().float(); w = w.contiguous().float()
  B, C, H, W = x.shape
  kh, kw = w.shape[-2:]
  out = torch.empty((B, 1, H - kh + 1, W - kw + 1), dtype=torch.float32, device=x.device)
  with torch.cuda.device(x.device):
    rc = load().srl_xcorr_forward(x.data_ptr(), w.data_ptr(), out.data_ptr(), B, C, H, W, kh, kw, _stream(x))
  if rc:
    raise RuntimeError(load().srl_qnet_last_error().decode())
  return out


def policy_head(adv, u, rnd, epsilon):
  """Epsilon-greedy head (dqn.py:334-348): adv [B,A] f32, u [B] f32, rnd [B] i64 -> actions [B] i64."""
  adv = adv.contiguous().float()
  B, A = adv.shape
  actions = torch.empty(B, dtype=torch.int64, device=adv.device)
  with torch.cuda.device(adv.device):
    rc = load().srl_policy_head(adv.data_ptr(), u.contiguous().data_ptr(), rnd.contiguous().data_ptr(),
                                float(epsilon), actions.data_ptr(), B, A, _stream(adv))
  if rc:
    raise RuntimeError(load().srl_qnet_last_error().decode())
  return actions


_CL = torch.channels_last


def _cl(t):
  return t if t.is_contiguous(memory_format=_CL) else t.contiguous(memory_format=_CL)


def bias_act(y, bias, out=None, out_offset=0, relu=True, nchw=False):
  """relu(y + bias[c]) in one pass (csrc/epilogue.hip).  y: bf16 or float32 [B,C,H,W] channels-last.  out: None = in
  place; a channels-last tensor of the same dtype with >= C channels = write into its channel slice
  [out_offset, out_offset + C); nchw=True returns a new NCHW-contiguous tensor (the layout the cross-correlation kernel
  reads)."""
  B, C, H, W = y.shape
  if nchw:
    dst = torch.empty((B, C, H, W), dtype=y.dtype, device=y.device)
    stride, hw = C, H * W
  else:
    dst = y if out is None else out
    stride, hw = dst.shape[1], 0
  fn = load().srl_bias_act_f32 if y.dtype == torch.float32 else load().srl_bias_act
  with torch.cuda.device(y.device):
    rc = fn(y.data_ptr(), dst.data_ptr(), bias.data_ptr(), B * H * W, C, stride, out_offset, hw, int(relu), _stream(y))
  if rc:
    raise RuntimeError(load().srl_epilogue_last_error().decode())
  return dst


class _BiasAct(torch.autograd.Function):
  """y <- relu(y + bias) in place on the convolution's output, with the hand-written backward (csrc/epilogue.hip
  k_bias_act_bwd): gx = gy * (y > 0) and the bias gradient summed in a fixed order, one pass over gy and y."""

  @staticmethod
  def forward(ctx, y, bias, relu):
    bias_act(y, bias, relu=relu)
    ctx.mark_dirty(y)
    ctx.relu = bool(relu)
    ctx.save_for_backward(y)
    return y

  @staticmethod
  def backward(ctx, gy):
    y, = ctx.saved_tensors
    B, C, H, W = y.shape
    gy = gy.contiguous(memory_format=_CL)
    gx = torch.empty_like(gy, memory_format=_CL)
    gb = torch.empty(C, dtype=torch.float32, device=y.device)
    L = load()
    scratch = torch.empty(L.srl_bias_act_bwd_scratch_floats(B * H * W, C), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
      rc = L.srl_bias_act_bwd_f32(gy.data_ptr(), y.data_ptr(), gx.data_ptr(), gb.data_ptr(), scratch.data_ptr(), B * H * W, C,
                                  int(ctx.relu), _stream(y))
    if rc:
      raise RuntimeError(L.srl_epilogue_last_error().decode())
    return gx, gb, None


def bias_act_supported(channels):
  return channels % 8 == 0 and channels <= 256 and 256 % (channels // 8) == 0


def bias_act_autograd(y, bias, relu=True):
  """relu(y + bias[c]) (or y + bias[c]) of a float32 convolution output [B,C,H,W], differentiable: in place on y (made
  channels-last), forward and backward by the kernels of csrc/epilogue.hip.  The update path's replacement for the
  library's separate bias add, ReLU, ReLU backward and bias-gradient reduction."""
  if y.dtype != torch.float32 or not y.is_cuda:
    raise RuntimeError('bias_act_autograd: float32 CUDA tensors only')
  y = y.contiguous(memory_format=_CL)
  if not (torch.is_grad_enabled() and (y.requires_grad or bias.requires_grad)):
    return bias_act(y, bias.detach(), relu=relu)
  return _BiasAct.apply(y, bias, relu)


def bias_act_pool(y, bias, skip, skip_offset):
  """relu(y + bias[c]) into the channel slice [skip_offset, skip_offset + C) of `skip`, plus its 2 x 2 max-pool."""
  B, C, H, W = y.shape
  pooled = torch.empty((B, C, H // 2, W // 2), dtype=y.dtype, device=y.device, memory_format=_CL)
  fn = load().srl_bias_act_pool_f32 if y.dtype == torch.float32 else load().srl_bias_act_pool
  with torch.cuda.device(y.device):
    rc = fn(y.data_ptr(), skip.data_ptr(), pooled.data_ptr(), bias.data_ptr(), B, H, W, C, skip.shape[1], skip_offset, _stream(y))
  if rc:
    raise RuntimeError(load().srl_epilogue_last_error().decode())
  return pooled


def pool2x2(buf, C, offset=0):
  """2 x 2 max-pool of the channel slice [offset, offset + C) of a channels-last bf16 / float32 buffer [B,Ctot,H,W]."""
  B, Ct, H, W = buf.shape
  pooled = torch.empty((B, C, H // 2, W // 2), dtype=buf.dtype, device=buf.device, memory_format=_CL)
  with torch.cuda.device(buf.device):
    rc = load().srl_pool2x2(buf.data_ptr(), pooled.data_ptr(), B, H, W, C, Ct, offset, int(buf.dtype == torch.float32), _stream(buf))
  if rc:
    raise RuntimeError(load().srl_epilogue_last_error().decode())
  return pooled


_PACK_INDEX = {}


def _pack(w, kind, build, x3=False):
  """Gather of a weight tensor into a kernel's fragment order.  The flat gather index (and the mask of padded taps)
  depends on the layer shape only and is built once per (kind, shape, device): re-packing after every weight update —
  once per rollout while training — is then one gather per layer.  x3: the fragments of bf16(w), then those of
  bf16(w - bf16(w)) (the fp32-class kernels)."""
  key = (kind, tuple(w.shape), w.device)
  if key not in _PACK_INDEX:
    _PACK_INDEX[key] = build()
  idx, mask = _PACK_INDEX[key]
  flat = w.detach().float().reshape(-1)
  if x3:
    hi = flat.to(torch.bfloat16)
    lo = (flat - hi.float()).to(torch.bfloat16)
    out = torch.cat([hi[idx], lo[idx]])
    return out * torch.cat([mask, mask]) if mask is not None else out
  out = flat.to(torch.bfloat16)[idx]
  return out * mask if mask is not None else out


def _conv3x3_index(cout, cin, dev):
  ks = torch.arange(5 if cin == 16 else 9 * (cin // 32), device=dev)[:, None, None, None]
  mt = torch.arange(cout // 16, device=dev)[None, :, None, None]
  lane = torch.arange(64, device=dev)[None, None, :, None]
  j = torch.arange(8, device=dev)[None, None, None, :]
  k = 8 * (lane >> 4) + j
  if cin == 16:
    tap, ci = 2 * ks + (k >> 4), k & 15
  else:
    m = cin // 32
    tap, ci = ks // m + 0 * k, 32 * (ks % m) + k
  co = 16 * mt + (lane & 15)
  tapc = tap.clamp(max=8)
  idx = ((co * cin + ci) * 9 + tapc).reshape(-1)
  mask = (tap < 9).expand_as(idx.reshape(tapc.shape[0], co.shape[1], 64, 8)).reshape(-1).to(torch.bfloat16) if cin == 16 else None
  return idx, mask


def pack_conv3x3_weights(w, x3=False):
  """Conv2d weight [cout, cin, 3, 3] -> bf16 A-fragment order of csrc/conv_mfma.hip (see include/stackrl_qnet.h)."""
  cout, cin = int(w.shape[0]), int(w.shape[1])
  n = load().srl_conv3x3_wfrag_elems(cin, cout)
  if n < 0:
    raise ValueError('conv3x3 MFMA kernel: unsupported channels %d -> %d' % (cin, cout))
  out = _pack(w, 'conv3x3', lambda: _conv3x3_index(cout, cin, w.device), x3)
  assert out.numel() == n * (2 if x3 else 1)
  return out


def pack_conv3x3_weights_x3(w):
  """Conv2d weight [cout, cin, 3, 3] (cin in {16, 32, 64}, cout in {16, 32}) -> the two bf16 fragment sets of the fp32-class kernel
  (srl_conv3x3_bias_relu_f32): the fragments of bf16(w), then those of bf16(w - bf16(w))."""
  return pack_conv3x3_weights(w, x3=True)


def conv3x3_bias_relu(x, wfrag, bias, cout, out=None, out_offset=0, pool=False, nchw=False):
  """relu(conv3x3(x) + bias) on the matrix cores (csrc/conv_mfma.hip).  x: bf16 [B,cin,H,W] channels-last.  Returns
  the output (a new channels-last tensor, or `out` whose channel slice [out_offset, out_offset + cout) was written, or
  an NCHW-contiguous tensor with nchw=True) and, with pool=True, also the 2 x 2 max-pooled tensor."""
  B, cin, H, W = x.shape
  if nchw:
    dst = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device)
    stride = cout
  else:
    dst = out if out is not None else torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=_CL)
    stride = dst.shape[1]
  pooled = torch.empty((B, cout, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=_CL) if pool else None
  # float32 tensors take the fp32-class kernel (bf16x3 products, wfrag from pack_conv3x3_weights_x3)
  fn = load().srl_conv3x3_bias_relu_f32 if x.dtype == torch.float32 else load().srl_conv3x3_bias_relu
  with torch.cuda.device(x.device):
    rc = fn(x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), dst.data_ptr(), pooled.data_ptr() if pool else None, B, H, W, cin,
            cout, stride, out_offset, int(nchw), _stream(x))
  if rc:
    raise RuntimeError(load().srl_conv_last_error().decode())
  return (dst, pooled) if pool else dst


def _conv3x3_gemm_index(cout, cin, dev):
  cb = torch.arange(cin // 32, device=dev)[:, None, None, None, None]
  tap = torch.arange(9, device=dev)[None, :, None, None, None]
  mt = torch.arange(cout // 16, device=dev)[None, None, :, None, None]
  lane = torch.arange(64, device=dev)[None, None, None, :, None]
  j = torch.arange(8, device=dev)[None, None, None, None, :]
  co, ci = 16 * mt + (lane & 15), 32 * cb + 8 * (lane >> 4) + j
  return ((co * cin + ci) * 9 + tap).reshape(-1), None


def pack_conv3x3_gemm_weights(w, x3=False):
  """Conv2d weight [cout, cin, 3, 3] -> bf16 A-fragment order of csrc/conv_gemm.hip ([cin / 32][tap][cout / 16][lane][8]);
  x3: the fragments of bf16(w) followed by those of bf16(w - bf16(w)) (the fp32-class kernel)."""
  cout, cin = int(w.shape[0]), int(w.shape[1])
  out = _pack(w, 'conv3x3_gemm', lambda: _conv3x3_gemm_index(cout, cin, w.device), x3)
  assert out.numel() == load().srl_conv3x3_gemm_wfrag_elems(cin, cout) * (2 if x3 else 1)
  return out


def conv3x3_gemm_supported(cin, cout, W, B):
  L = load()
  return bool(L.srl_conv3x3_gemm_supported(cin, cout, W)) and B % L.srl_conv3x3_gemm_batch_multiple(cout, W) == 0


def conv3x3_gemm_bias_relu(x, wfrag, bias, cout, out=None, out_offset=0):
  """relu(conv3x3(x) + bias) of a deep U-Net level as an implicit GEMM on the matrix cores (csrc/conv_gemm.hip).
  x: bf16 or float32 (fp32-class products, wfrag packed with x3=True) [B,cin,W,W] channels-last; returns a new
  channels-last tensor or `out` whose channel slice [out_offset, out_offset + cout) was written."""
  B, cin, H, W = x.shape
  assert H == W
  dst = out if out is not None else torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=_CL)
  with torch.cuda.device(x.device):
    rc = load().srl_conv3x3_gemm_bias_relu(x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), dst.data_ptr(), B, W, cin, cout,
                                           dst.shape[1], out_offset, int(x.dtype == torch.float32), _stream(x))
  if rc:
    raise RuntimeError(load().srl_conv_gemm_last_error().decode())
  return dst


def _convt2x2_index(cin, cout, dev):
  ks = torch.arange(cin // 32, device=dev)[:, None, None, None]
  mt = torch.arange(4 * cout // 16, device=dev)[None, :, None, None]
  lane = torch.arange(64, device=dev)[None, None, :, None]
  j = torch.arange(8, device=dev)[None, None, None, :]
  ci = 32 * ks + 8 * (lane >> 4) + j
  m = 16 * mt + (lane & 15)
  q, co = m // cout, m % cout
  return ((ci * cout + co) * 4 + q).reshape(-1), None     # w[ci][co][q >> 1][q & 1]


def pack_convt2x2_weights(w, x3=False):
  """ConvTranspose2d weight [cin, cout, 2, 2] -> bf16 A-fragment order of k_convt2x2 (include/stackrl_qnet.h)."""
  cin, cout = int(w.shape[0]), int(w.shape[1])
  n = load().srl_convt2x2_wfrag_elems(cin, cout)
  if n < 0:
    raise ValueError('convT 2x2 MFMA kernel: unsupported channels %d -> %d' % (cin, cout))
  out = _pack(w, 'convt2x2', lambda: _convt2x2_index(cin, cout, w.device), x3)
  assert out.numel() == n * (2 if x3 else 1)
  return out


def pack_convt2x2_weights_x3(w):
  """ConvTranspose2d weight -> the hi and lo bf16 fragment sets of the fp32-class kernel (srl_convt2x2_bias_relu_f32)."""
  return pack_convt2x2_weights(w, x3=True)


def convt2x2_gemm_bias_relu(x, wfrag, bias, cout, out, out_offset=0):
  """relu(conv_transpose2d(x, k=2, s=2) + bias) of the deep levels (128 -> 64, 256 -> 128; any map size) into the channel
  slice of `out`, csrc/conv_gemm.hip k_convt2x2_gemm; bf16 or float32 (wfrag from pack_convt2x2_weights(w, x3=True))."""
  B, cin, H, W = x.shape
  with torch.cuda.device(x.device):
    rc = load().srl_convt2x2_gemm_bias_relu(x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, cin, cout,
                                            out.shape[1], out_offset, int(x.dtype == torch.float32), _stream(x))
  if rc:
    raise RuntimeError(load().srl_conv_gemm_last_error().decode())
  return out


def convt2x2_bias_relu(x, wfrag, bias, cout, out, out_offset=0):
  """relu(conv_transpose2d(x, k=2, s=2) + bias) into the channel slice [out_offset, out_offset + cout) of `out`
  (channels-last, twice the spatial size of x), csrc/conv_mfma.hip.  bf16 tensors: bf16 MFMA; float32 tensors: the
  fp32-class kernel (wfrag from pack_convt2x2_weights_x3)."""
  B, cin, H, W = x.shape
  fn = load().srl_convt2x2_bias_relu_f32 if x.dtype == torch.float32 else load().srl_convt2x2_bias_relu
  with torch.cuda.device(x.device):
    rc = fn(x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, cin, cout,
                                       out.shape[1], out_offset, _stream(x))
  if rc:
    raise RuntimeError(load().srl_conv_last_error().decode())
  return out


def pack_thin_weights(w):
  """Conv2d weight [16, cin, 3, 3] (cin 1 or 2) -> float32 [16, 3, 3, cin], the order the thin-layer kernels read."""
  return w.detach().float().permute(0, 2, 3, 1).contiguous()


def conv3x3_thin(x, w, bias, out=None, dtype=torch.bfloat16):
  """relu(conv3x3(x) + bias) for 1 or 2 input channels -> 16 (csrc/conv_mfma.hip, vector ALU).  x: uint8 (scaled by
  1/255) or float32, channels-last memory [B,H,W,cin]; w from `pack_thin_weights`.  Returns bf16 (or float32, per `dtype` / `out`) [B,16,Hp,Wp]
  channels-last; `out` (optional) is a larger zero-margined buffer of that kind whose top-left H x W region is written."""
  B, H, W, cin = x.shape
  if out is None:
    out = torch.empty((B, 16, H, W), dtype=dtype, device=x.device, memory_format=_CL)
  fn = load().srl_conv3x3_thin_f32 if out.dtype == torch.float32 else load().srl_conv3x3_thin
  with torch.cuda.device(x.device):
    rc = fn(x.data_ptr(), int(x.dtype != torch.uint8), w.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                 B, H, W, cin, out.shape[2], out.shape[3], _stream(x))
  if rc:
    raise RuntimeError(load().srl_conv_last_error().decode())
  return out


def conv3x3_relu_project(x, wfrag, bias, proj_w, proj_b, hv, wv):
  """sum_c proj_w[c] relu(conv3x3(x)[c] + bias[c]) + proj_b, float32 [B,hv,wv]: the last two layers of `pos_layers`."""
  B, _, H, W = x.shape
  out = torch.empty((B, hv, wv), dtype=torch.float32, device=x.device)
  fn = load().srl_conv3x3_relu_project_f32 if x.dtype == torch.float32 else load().srl_conv3x3_relu_project
  with torch.cuda.device(x.device):
    rc = fn(x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), proj_w.data_ptr(), float(proj_b),
                                         out.data_ptr(), B, H, W, hv, wv, _stream(x))
  if rc:
    raise RuntimeError(load().srl_conv_last_error().decode())
  return out


def thin_conv3x3_bias_relu(x, w1, b1, wfrag, bias, out=None, out_offset=0, pool=False, nchw=False):
  """`conv3x3_thin` (float32 output) followed by `conv3x3_bias_relu` (16 -> 16, fp32-class) as one kernel: the 16-channel
  intermediate stays in LDS.  x: uint8 or float32 [B,H,W,cin] (cin 1 or 2; H, W multiples of 16), w1 from
  `pack_thin_weights`; returns what
  `conv3x3_bias_relu` returns, equal bit for bit."""
  B, H, W, cin = x.shape
  if nchw:
    dst = torch.empty((B, 16, H, W), dtype=torch.float32, device=x.device)
    stride = 16
  else:
    dst = out if out is not None else torch.empty((B, 16, H, W), dtype=torch.float32, device=x.device, memory_format=_CL)
    stride = dst.shape[1]
  pooled = torch.empty((B, 16, H // 2, W // 2), dtype=torch.float32, device=x.device, memory_format=_CL) if pool else None
  with torch.cuda.device(x.device):
    rc = load().srl_thin_conv3x3_bias_relu_f32(x.data_ptr(), int(x.dtype != torch.uint8), cin, w1.data_ptr(), b1.data_ptr(),
                                               wfrag.data_ptr(), bias.data_ptr(), dst.data_ptr(), pooled.data_ptr() if pool else None,
                                               B, H, W, stride, out_offset, int(nchw), _stream(x))
  if rc:
    raise RuntimeError(load().srl_conv_last_error().decode())
  return (dst, pooled) if pool else dst


def thin_conv3x3_relu_project(x, w1, b1, wfrag, bias, proj_w, proj_b):
  """`pos_layers` whole as one kernel (fp32-class): x float32 [B,H,W] -> float32 [B,H,W]; equal bit for bit to
  `conv3x3_thin` into a zero-margined map followed by `conv3x3_relu_project`."""
  B, H, W = x.shape
  out = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
  with torch.cuda.device(x.device):
    rc = load().srl_thin_conv3x3_relu_project_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), wfrag.data_ptr(), bias.data_ptr(),
                                                  proj_w.data_ptr(), float(proj_b), out.data_ptr(), B, H, W, _stream(x))
  if rc:
    raise RuntimeError(load().srl_conv_last_error().decode())
  return out


class _LazyPacked(object):
  """Packed weights by module: `m in d` = the module has a hand-written kernel; `d[m]` packs on first use after a weight
  update (a layer whose kernel is not used at the current map size is never packed)."""

  def __init__(self):
    self._make = {}
    self._done = {}

  def offer(self, m, make):
    self._make[m] = make

  def __contains__(self, m):
    return m in self._make

  def __getitem__(self, m):
    if m not in self._done:
      self._done[m] = self._make[m]()
    return self._done[m]

  def packed(self):
    """(module, packed weights) of what has been packed since the last weight update."""
    return list(self._done.items())


class _WeightBias(object):
  """(weight in the forward's dtype and channels-last, float32 bias) of a module; the weight copy is made on first use."""

  def __init__(self, m, dtype):
    self._m, self._dtype, self._w = m, dtype, None
    self._b = m.bias.detach().float().contiguous()

  def __getitem__(self, i):
    if i == 1:
      return self._b
    if i != 0:
      raise IndexError(i)
    if self._w is None:
      self._w = self._m.weight.detach().to(self._dtype).contiguous(memory_format=_CL)
    return self._w

  def __iter__(self):
    return iter((self[0], self[1]))


class FastFeatures(object):
  """Inference-only forward of the two U-Nets (`DeepQSiamFCN.features`, models.py:160-177; `layers.unet`,
  layers.py:135-259) in bf16 channels-last: library convolutions without bias, and the fused element-wise passes of
  csrc/epilogue.hip instead of separate bias / ReLU / max-pool / concatenate / layout kernels.  Returns the left and
  right feature maps NCHW-contiguous, ready for the MFMA cross-correlation."""

  def __init__(self, net, mfma_conv=True, dtype=torch.bfloat16, x3_conv=True, fuse_thin=True):
    self.net = net
    self.fuse_thin = bool(fuse_thin)   # fp32-class: thin layer + the 16 -> 16 layer behind it as one kernel (False: two, same values)
    # dtype float32 = the reference's dtype: (x3_conv) the same hand-written layers as the bf16 mode, in fp32-class
    # precision — the 16- / 32-output-channel 3 x 3 layers, the 32 -> 16 / 64 -> 32 transposed convolutions and the
    # position head with bf16x3 products on the matrix cores (csrc/conv_mfma.hip k_conv3x3_x3, k_convt2x2_x3), the thin
    # first layers in fp32 on the vector ALU; the >= 64-output-channel layers are library fp32 convolutions without bias
    # + the fused fp32 epilogues
    self.dtype = dtype
    self.mfma_conv = bool(mfma_conv) and dtype == torch.bfloat16   # hand-written MFMA kernel for the 16- / 32-channel 3 x 3 layers
    self.x3_conv = bool(x3_conv) and dtype == torch.float32
    self._key = None
    self._w = {}
    self._wf = _LazyPacked()
    self._wt = {}
    self._wg = _LazyPacked()
    self._w1 = _LazyPacked()
    self._pos = None
    self._posbuf = {}

  def _refresh(self):
    # `_weights_epoch` is bumped by whoever updates the parameters outside ATen's sight: a hipGraph replay of the
    # optimiser step (DQN._train_graphed) changes the weights without touching any tensor's version counter
    key = tuple(p._version for p in self.net.parameters()) + (id(self.net), getattr(self.net, '_weights_epoch', 0))
    if key == self._key:
      return
    self._w = {}
    self._wf = _LazyPacked()
    self._wt = {}
    self._wg = _LazyPacked()
    self._w1 = _LazyPacked()        # transposed convolutions left to the float32 1 x 1 form of the update's kernel (srl_tconv)
    for m in self.net.modules():
      if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
        self._w[m] = _WeightBias(m, self.dtype)
        if self.mfma_conv and isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and \
           m.in_channels in (16, 32, 64) and m.out_channels in (16, 32):
          self._wf.offer(m, lambda m=m: pack_conv3x3_weights(m.weight))
        if self.x3_conv and isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and \
           m.in_channels in (16, 32, 64) and m.out_channels in (16, 32):
          self._wf.offer(m, lambda m=m: pack_conv3x3_weights_x3(m.weight))
        # the deep levels (64 / 128 / 256 output channels): implicit GEMM on the matrix cores (csrc/conv_gemm.hip)
        if (self.mfma_conv or self.x3_conv) and isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and \
           m.out_channels in (64, 128, 256) and m.in_channels % 32 == 0:
          self._wg.offer(m, lambda m=m: pack_conv3x3_gemm_weights(m.weight, x3=self.x3_conv))
        if (self.mfma_conv or self.x3_conv) and isinstance(m, torch.nn.ConvTranspose2d) and m.kernel_size == (2, 2) and \
           (m.in_channels, m.out_channels) in ((32, 16), (64, 32)):
          self._wf.offer(m, lambda m=m: pack_convt2x2_weights(m.weight, x3=not self.mfma_conv))
        if (self.mfma_conv or self.x3_conv) and isinstance(m, torch.nn.ConvTranspose2d) and m.kernel_size == (2, 2) and \
           (m.in_channels, m.out_channels) in ((128, 64), (256, 128)):
          self._wg.offer(m, lambda m=m: pack_convt2x2_weights(m.weight, x3=not self.mfma_conv))
        # float32 rollout: any other 2 x 2 stride-2 transposed convolution (the right U-Net's 64 -> 32 at 8 x 8, whose rows
        # are narrower than the MFMA kernel's 16-pixel tiles) as a 1 x 1 convolution to 4 cout channels + depth-to-space in
        # true float32 on the matrix cores (csrc/train_conv.hip k_tconv): weights [cin][q cout + co], q = 2 dy + dx
        if self.x3_conv and isinstance(m, torch.nn.ConvTranspose2d) and m.kernel_size == (2, 2) and m.stride == (2, 2) and \
           m.out_channels % 4 == 0 and (4 * m.out_channels) % 16 == 0:
          self._w1.offer(m, lambda m=m: m.weight.detach().float().permute(0, 2, 3, 1).reshape(m.in_channels, 4 * m.out_channels).contiguous())
        if (self.mfma_conv or self.x3_conv) and isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and \
           m.in_channels in (1, 2) and m.out_channels == 16:
          self._wt[m] = pack_thin_weights(m.weight)
    pos = getattr(self.net, 'pos', None)
    self._pos = None
    if (self.mfma_conv or self.x3_conv) and pos is not None and len(pos) == 5 and pos[0] in self._wt and pos[2] in self._wf and \
       pos[4].kernel_size == (1, 1) and pos[4].in_channels == 16 and pos[4].out_channels == 1:
      self._pos = (pos[4].weight.detach().float().reshape(16).contiguous(), float(pos[4].bias.detach()))
    self._key = key

  def _mine(self, m, x):
    return m in self._wf and x.shape[2] % 16 == 0 and x.shape[3] % 16 == 0

  def _gemm(self, m, x):
    return m in self._wg and x.shape[2] == x.shape[3] and \
      conv3x3_gemm_supported(m.in_channels, m.out_channels, x.shape[3], x.shape[0])

  def _conv(self, m, x):
    w, b = self._w[m]
    return _cl(torch.nn.functional.conv2d(x, w, None, padding=m.padding)), b

  def _unet(self, U, obs):
    """obs: the env's uint8 observation [B,H,W,c] (channels-last memory)."""
    F = torch.nn.functional
    B = obs.shape[0]
    cats = []
    x = None
    for blk in U.down:
      f = blk[0].out_channels
      if x is None and self.fuse_thin and self.x3_conv and blk[0] in self._wt and blk[2] in self._wf and f == 16 and \
         blk[2].in_channels == 16 and obs.shape[1] % 16 == 0 and obs.shape[2] % 16 == 0:
        cat = torch.empty((B, 2 * f, obs.shape[1], obs.shape[2]), dtype=self.dtype, device=obs.device, memory_format=_CL)
        _, x = thin_conv3x3_bias_relu(obs, self._wt[blk[0]], self._w[blk[0]][1], self._wf[blk[2]], self._w[blk[2]][1],
                                      out=cat, out_offset=f, pool=True)
        cats.append(cat)
        continue
      if x is None and blk[0] in self._wt:
        y = conv3x3_thin(obs, self._wt[blk[0]], self._w[blk[0]][1], dtype=self.dtype)   # /255 and the cast happen in the kernel
      elif x is None:
        # uint8 NHWC / 255 (models.py:144-147); the NHWC memory is exactly a channels-last NCHW tensor
        x = (obs.float() / 255.0).to(self.dtype).permute(0, 3, 1, 2)
        y, b = self._conv(blk[0], x)
        bias_act(y, b)
      elif self._mine(blk[0], x):
        y = conv3x3_bias_relu(x, self._wf[blk[0]], self._w[blk[0]][1], f)
      elif self._gemm(blk[0], x):
        y = conv3x3_gemm_bias_relu(x, self._wg[blk[0]], self._w[blk[0]][1], f)
      else:
        y, b = self._conv(blk[0], x)
        bias_act(y, b)
      cat = torch.empty((B, 2 * f, y.shape[2], y.shape[3]), dtype=y.dtype, device=y.device, memory_format=_CL)
      if self._mine(blk[2], y):                     # skip -> second half of the decoder's concat buffer, + pooled
        _, x = conv3x3_bias_relu(y, self._wf[blk[2]], self._w[blk[2]][1], f, out=cat, out_offset=f, pool=True)
      elif self._gemm(blk[2], y):
        conv3x3_gemm_bias_relu(y, self._wg[blk[2]], self._w[blk[2]][1], f, out=cat, out_offset=f)
        x = pool2x2(cat, f, f)
      else:
        y, b = self._conv(blk[2], y)
        x = bias_act_pool(y, b, cat, f)
      cats.append(cat)
    for m in (U.bottom[0], U.bottom[2]):
      if self._gemm(m, x):
        x = conv3x3_gemm_bias_relu(x, self._wg[m], self._w[m][1], m.out_channels)
      else:
        y, b = self._conv(m, x)
        x = bias_act(y, b)
    n = len(U.up)
    for k, (up, blk) in enumerate(zip(U.up, U.upconv)):
      cat = cats.pop()
      f = up.out_channels
      b = self._w[up][1]
      if up in self._wf and x.shape[3] % 16 == 0 and x.is_contiguous(memory_format=_CL):
        convt2x2_bias_relu(x, self._wf[up], b, f, cat, 0)
      elif up in self._wg and x.is_contiguous(memory_format=_CL):
        convt2x2_gemm_bias_relu(x, self._wg[up], b, f, cat, 0)
      elif up in self._w1 and x.dtype == torch.float32 and x.is_contiguous(memory_format=_CL) and cat.is_contiguous(memory_format=_CL):
        from stackrl_amd import qtrain                    # channels-last tensors are [B, H, W, C] in memory
        qtrain.tconv(qtrain.Act(x.permute(0, 2, 3, 1)), self._w1[up], b, 4 * f, taps=1, relu=True,
                     out=(cat.permute(0, 2, 3, 1), 0), d2s=f)
      else:
        y = _cl(F.conv_transpose2d(x, self._w[up][0], None, stride=up.stride))
        bias_act(y, b, out=cat, out_offset=0)       # Concatenate([up, skip]) without a copy
      if self._mine(blk[0], cat):
        y = conv3x3_bias_relu(cat, self._wf[blk[0]], self._w[blk[0]][1], f)
      elif self._gemm(blk[0], cat):
        y = conv3x3_gemm_bias_relu(cat, self._wg[blk[0]], self._w[blk[0]][1], f)
      else:
        y, b = self._conv(blk[0], cat)
        bias_act(y, b)
      last = k == n - 1
      if self._mine(blk[2], y):
        x = conv3x3_bias_relu(y, self._wf[blk[2]], self._w[blk[2]][1], f, nchw=last)
      elif self._gemm(blk[2], y) and not last:
        x = conv3x3_gemm_bias_relu(y, self._wg[blk[2]], self._w[blk[2]][1], f)
      else:
        y, b = self._conv(blk[2], y)
        x = bias_act(y, b, nchw=last)
    return x

  @torch.no_grad()
  def __call__(self, inputs):
    self._refresh()
    xm, xo = inputs
    return self._unet(self.net.left, xm.contiguous()), self._unet(self.net.right, xo.contiguous())

  @torch.no_grad()
  def pos(self, corr):
    """`pos_layers` (layers.py:439-472) on the correlation map [B,1,oh,ow] float32 -> advantages [B, oh*ow] float32:
    the 1 -> 16 layer on the vector ALU into a zero-margined map (of this object's dtype) padded to a multiple of 16, then
    the 16 -> 16 layer on the matrix cores with the final 1 x 1 projection fused into its epilogue (fp32 from the
    accumulators)."""
    self._refresh()
    if self._pos is None:
      return self.net.pos(corr).flatten(1)
    pos = self.net.pos
    B, _, oh, ow = corr.shape
    pw, pb = self._pos
    if self.fuse_thin and self.x3_conv:       # fp32-class: the three layers as one kernel, the 16-channel maps never leave LDS
      return thin_conv3x3_relu_project(corr.reshape(B, oh, ow).contiguous(), self._wt[pos[0]], self._w[pos[0]][1], self._wf[pos[2]],
                                       self._w[pos[2]][1], pw, pb).flatten(1)
    hp, wp = (oh + 15) // 16 * 16, (ow + 15) // 16 * 16
    key = (B, hp, wp, corr.device.index)
    buf = self._posbuf.get(key)
    if buf is None:
      buf = torch.zeros((B, 16, hp, wp), dtype=self.dtype, device=corr.device).contiguous(memory_format=_CL)
      self._posbuf = {key: buf}
    conv3x3_thin(corr.reshape(B, oh, ow, 1).contiguous(), self._wt[pos[0]], self._w[pos[0]][1], out=buf)
    return conv3x3_relu_project(buf, self._wf[pos[2]], self._w[pos[2]][1], pw, pb, oh, ow).flatten(1)


class FusedPolicy(object):
  """Rollout policy (`DQN.collect` -> `policy(exploration=True)`, dqn.py:391-395) with the hand-written head:
  library convs for the two U-Nets and the position convs, HIP cross-correlation, HIP arg-max + epsilon-greedy.
  Draws the same random numbers in the same order as `DQN.policy`, so both paths give identical actions."""

  def __init__(self, chunk=2048, autocast=None, fast=None):
    # rollout batches are processed in chunks to bound activation memory: 2,048 samples hold ~10 GB of fp32 activations
    # (of 288 GB) and run 4 % faster per sample than 512 (more workgroups per launch: 24.9 against 26.1 ms per 4,096, bf16)
    self.chunk = int(chunk)
    self.autocast = autocast     # None = fp32 like the reference; torch.bfloat16 runs the library convs on MFMA
    # fused epilogues around bias-free library convolutions (bf16: + the MFMA convolution kernels; fp32: epilogues only)
    self.fast = (autocast == torch.bfloat16) if fast is None else bool(fast)
    self._ff = None

  @staticmethod
  def draws(net, B, gen, device):
    """The random numbers one call over B samples consumes, drawn as that call draws them (`draws=` of `__call__`: a
    policy evaluated group by group passes each group its slice and takes the actions of one call over the batch)."""
    u = torch.rand(B, generator=gen, device=device)
    return u, torch.randint(net.n_actions, (B,), generator=gen, device=device)

  @torch.no_grad()
  def __call__(self, net, inputs, epsilon, gen, draws=None):
    xm, xo = inputs
    B = xm.shape[0]
    u, rnd = draws if draws is not None else self.draws(net, B, gen, xm.device)
    out = torch.empty(B, dtype=torch.int64, device=xm.device)
    for s in range(0, B, self.chunk):
      e = min(B, s + self.chunk)
      if self.fast:
        if self._ff is None or self._ff.net is not net:
          self._ff = FastFeatures(net, dtype=torch.bfloat16 if self.autocast == torch.bfloat16 else torch.float32)
        x, w = self._ff((xm[s:e], xo[s:e]))
      elif self.autocast is not None:
        with torch.autocast('cuda', dtype=self.autocast):
          x, _, w = net.features((xm[s:e], xo[s:e]))
      else:
        x, _, w = net.features((xm[s:e], xo[s:e]))
      corr = xcorr_forward(x, w)
      adv = self._ff.pos(corr) if self.fast else net.pos(corr).flatten(1)
      out[s:e] = policy_head(adv, u[s:e], rnd[s:e], epsilon)
    return out


# ------------------------------------------------------------------------------------------------ update path (csrc/learner.hip)
def _lchk(rc):
  if rc:
    raise RuntimeError(load().srl_learner_last_error().decode())


def _ptr(t):
  return None if t is None else t.data_ptr()


def td_epilogue(q, q_next_online, q_next_target, actions, rewards, terminal, weights, gamma, huber_delta, reward_scale,
                double, prio_eps, ws):
  """Loss, mean TD, |TD|, new priorities and d loss / d Q(s, .) of `DQN.train` (dqn.py:408-476) in one launch.
  q, q_next_*: float32 [mb, A]; ws: a dict the caller keeps (scratch and the ticket word live at fixed addresses)."""
  mb, A = q.shape
  dev = q.device
  if 'ticket' not in ws or ws['mb'] != mb:
    ws.update(mb=mb, ticket=torch.zeros(1, dtype=torch.int32, device=dev), scratch=torch.empty(2 * mb, dtype=torch.float32, device=dev))
  out = torch.empty(2, dtype=torch.float32, device=dev)
  td_abs = torch.empty(mb, dtype=torch.float32, device=dev)
  logits = torch.empty(mb, dtype=torch.float32, device=dev)
  grad_q = torch.empty((mb, A), dtype=torch.float32, device=dev)
  q = q.contiguous(); qt = q_next_target.contiguous()
  qo = q_next_online.contiguous() if q_next_online is not None else None
  term = terminal.contiguous().view(torch.uint8)
  act = actions.contiguous(); rew = rewards.contiguous().float()
  wts = weights.contiguous().float() if weights is not None else None
  with torch.cuda.device(dev):
    _lchk(load().srl_td_epilogue(q.data_ptr(), _ptr(qo), qt.data_ptr(), act.data_ptr(), rew.data_ptr(), term.data_ptr(),
                                 _ptr(wts), float(gamma),
                                 -1.0 if huber_delta is None else float(huber_delta), float(reward_scale or 0.0), int(bool(double)),
                                 float(prio_eps), mb, A, out.data_ptr(), out[1:].data_ptr(), td_abs.data_ptr(), logits.data_ptr(),
                                 grad_q.data_ptr(), ws['scratch'].data_ptr(), ws['ticket'].data_ptr(), _stream(q)))
  return out[0], out[1], td_abs, logits, grad_q


def adam_step(params, grads, m, v, state, lr, beta1, beta2, eps):
  """Keras Adam over flat fp32 buckets, in place; `state` = 4 device floats {t, beta1^t, beta2^t, lr_t}."""
  with torch.cuda.device(params.device):
    _lchk(load().srl_adam_step(params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), params.numel(), state.data_ptr(),
                               float(lr), float(beta1), float(beta2), float(eps), _stream(params)))


def gumbel_topk(logits, u, alpha_t, k, ws):
  """K7: indices (int64 [k], descending key) and keys of the k largest alpha * logit + Gumbel(u)."""
  n = logits.numel()
  need = load().srl_gumbel_topk_scratch_bytes(n, k)
  if ws.get('topk_n') != (n, k):
    ws['topk_n'] = (n, k)
    ws['topk_scratch'] = torch.empty(need, dtype=torch.uint8, device=logits.device)
  idx = torch.empty(k, dtype=torch.int64, device=logits.device)
  key = torch.empty(k, dtype=torch.float32, device=logits.device)
  with torch.cuda.device(logits.device):
    _lchk(load().srl_gumbel_topk(logits.data_ptr(), u.data_ptr(), alpha_t.data_ptr(), n, k, idx.data_ptr(), key.data_ptr(),
                                 ws['topk_scratch'].data_ptr(), need, _stream(logits)))
  return idx, key


def replay_scatter(state, reward, terminal, action, slot, part_len, mem_states, mem_reward, mem_terminal, mem_action, mem_logits):
  """K8: one transition per env into row b * part_len + slot of the replay tensors (memory.py:153-161)."""
  s0, s1 = (t.contiguous() for t in state)
  B = s0.shape[0]
  b0, b1 = s0[0].numel() * s0.element_size(), s1[0].numel() * s1.element_size()
  r = reward.contiguous().float(); t = terminal.contiguous().to(torch.bool).view(torch.uint8); a = action.contiguous().to(torch.int64)
  with torch.cuda.device(s0.device):
    _lchk(load().srl_replay_scatter(s0.data_ptr(), s1.data_ptr(), b0, b1, r.data_ptr(), t.data_ptr(), a.data_ptr(), B, int(slot),
                                    int(part_len), mem_states[0].data_ptr(), mem_states[1].data_ptr(), mem_reward.data_ptr(),
                                    mem_terminal.data_ptr(), mem_action.data_ptr(), mem_logits.data_ptr(), _stream(s0)))


def logit_extrema(logits, ws):
  """(max logit, its lowest index), (min finite logit, its lowest index) as four 0-dim tensors (`ReplayMemory`'s tracker
  scans, memory.py:164-177, :282-316).  ws: a dict the caller keeps (scratch at a fixed address: hipGraph replay)."""
  dev = logits.device
  if 'ext' not in ws:
    ws['ext'] = torch.empty(int(load().srl_logit_extrema_scratch_bytes()), dtype=torch.uint8, device=dev)
  v = torch.empty(2, dtype=torch.float32, device=dev)
  i = torch.empty(2, dtype=torch.int64, device=dev)
  with torch.cuda.device(dev):
    _lchk(load().srl_logit_extrema(logits.data_ptr(), logits.numel(), v.data_ptr(), i.data_ptr(), ws['ext'].data_ptr(), _stream(logits)))
  return (v[0], i[0]), (v[1], i[1])


def replay_gather(idx, part_len, n_steps, literal_next, mem_states, mem_reward, mem_terminal, mem_action, mem_logits,
                  alpha_t=None, beta_t=None, min_logit=None):
  """K8: the minibatch for the sampled rows (memory.py:232-260): (states, actions, rewards, next_states, terminal), weights."""
  mb = idx.numel()
  m0, m1 = mem_states
  dev = m0.device
  b0, b1 = m0[0].numel() * m0.element_size(), m1[0].numel() * m1.element_size()
  s0 = torch.empty((mb,) + tuple(m0.shape[1:]), dtype=m0.dtype, device=dev); n0 = torch.empty_like(s0)
  s1 = torch.empty((mb,) + tuple(m1.shape[1:]), dtype=m1.dtype, device=dev); n1 = torch.empty_like(s1)
  act = torch.empty(mb, dtype=torch.int64, device=dev)
  rew = torch.empty(mb, dtype=torch.float32, device=dev)
  term = torch.empty(mb, dtype=torch.bool, device=dev)
  w = torch.empty(mb, dtype=torch.float32, device=dev) if alpha_t is not None else None
  with torch.cuda.device(dev):
    _lchk(load().srl_replay_gather(idx.data_ptr(), mb, int(part_len), int(n_steps), int(bool(literal_next)), None, m0.data_ptr(),
                                   m1.data_ptr(), b0, b1, mem_reward.data_ptr(), mem_terminal.data_ptr(), mem_action.data_ptr(),
                                   mem_logits.data_ptr(), _ptr(alpha_t), _ptr(beta_t), _ptr(min_logit), s0.data_ptr(), s1.data_ptr(),
                                   n0.data_ptr(), n1.data_ptr(), act.data_ptr(), rew.data_ptr(), term.data_ptr(), _ptr(w),
                                   _stream(m0)))
  return ((s0, s1), act, rew, (n0, n1), term), w
