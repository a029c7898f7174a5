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
    L.srl_policy_head.restype = ctypes.c_int
    L.srl_policy_head.argtypes = [VP, VP, VP, ctypes.c_float, VP, ctypes.c_int32, ctypes.c_int32, VP]
    L.srl_qnet_last_error.restype = ctypes.c_char_p
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


def xcorr_forward(x, w):
  """`layers.correlation` forward (layers.py:21-38): x [B,C,H,W], w [B,C,kh,kw] -> float32 [B,1,OH,OW].
  float32 features take the fp32 vector kernel (the reference's precision); bfloat16 features of the shapes in
  MFMA_SHAPES (the autocast rollout path) take the MFMA kernel."""
  if not x.is_cuda:
    raise RuntimeError('xcorr_forward needs a HIP device (no CPU fallback)')
  if x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and _mfma_ok(x, w):
    return xcorr_forward_mfma(x, w, BF16)
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


class FusedPolicy(object):
  """Rollout policy (`DQN.collect` -> `policy(exploration=True)`, dqn.py:391-395) with the hand-written head:
  library convs for the two U-Nets and the position convs, HIP cross-correlation, HIP arg-max + epsilon-greedy.
  Draws the same random numbers in the same order as `DQN.policy`, so both paths give identical actions."""

  def __init__(self, chunk=512, autocast=None):
    self.chunk = int(chunk)      # rollout batches are processed in chunks to bound activation memory
    self.autocast = autocast     # None = fp32 like the reference; torch.bfloat16 runs the library convs on MFMA

  @torch.no_grad()
  def __call__(self, net, inputs, epsilon, gen):
    xm, xo = inputs
    B = xm.shape[0]
    u = torch.rand(B, generator=gen, device=xm.device)
    rnd = torch.randint(net.n_actions, (B,), generator=gen, device=xm.device)
    out = torch.empty(B, dtype=torch.int64, device=xm.device)
    for s in range(0, B, self.chunk):
      e = min(B, s + self.chunk)
      if self.autocast is not None:
        with torch.autocast('cuda', dtype=self.autocast):
          x, _, w = net.features((xm[s:e], xo[s:e]))
      else:
        x, _, w = net.features((xm[s:e], xo[s:e]))
      adv = net.pos(xcorr_forward(x, w)).flatten(1)
      out[s:e] = policy_head(adv, u[s:e], rnd[s:e], epsilon)
    return out
