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
    L.srl_xcorr_bf16_scratch_bytes.restype = ctypes.c_int64
    L.srl_xcorr_bf16_scratch_bytes.argtypes = [ctypes.c_int32] * 3
    L.srl_xcorr_forward_bf16.restype = ctypes.c_int
    L.srl_xcorr_forward_bf16.argtypes = [VP, VP, VP, VP, ctypes.c_int64] + [ctypes.c_int32] * 6 + [VP]
    L.srl_xcorr_bf16_last_error.restype = ctypes.c_char_p
    L.srl_policy_head.restype = ctypes.c_int
    L.srl_policy_head.argtypes = [VP, VP, VP, ctypes.c_float, VP, ctypes.c_int32, ctypes.c_int32, VP]
    L.srl_qnet_last_error.restype = ctypes.c_char_p
    _LIB = L
  return _LIB


def _stream(t):
  return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


_SCRATCH = {}
MFMA_SHAPES = ((128, 32), (64, 16))   # (H, kh) instantiations of csrc/xcorr_mfma.hip


def xcorr_forward_bf16(x, w):
  """`layers.correlation` forward on the matrix cores: x [B,C,H,H], w [B,C,kh,kh] bfloat16 -> float32 [B,1,OH,OW]
  (bf16 products, fp32 accumulation)."""
  if not x.is_cuda:
    raise RuntimeError('xcorr_forward_bf16 needs a HIP device (no CPU fallback)')
  x = x.to(torch.bfloat16).contiguous(); w = w.to(torch.bfloat16).contiguous()
  B, C, H, W = x.shape
  kh, kw = w.shape[-2:]
  L = load()
  need = L.srl_xcorr_bf16_scratch_bytes(B, C, kh)
  if need < 0:
    raise RuntimeError('xcorr_forward_bf16: unsupported kernel size %d' % kh)
  key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
  scratch = _SCRATCH.get(key)
  if scratch is None or scratch.numel() < need:
    scratch = torch.empty(need, dtype=torch.uint8, device=x.device)
    _SCRATCH[key] = scratch
  out = torch.empty((B, 1, H - kh + 1, W - kw + 1), dtype=torch.float32, device=x.device)
  with torch.cuda.device(x.device):
    rc = L.srl_xcorr_forward_bf16(x.data_ptr(), w.data_ptr(), out.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                  B, C, H, W, kh, kw, _stream(x))
  if rc:
    raise RuntimeError(L.srl_xcorr_bf16_last_error().decode())
  return out


def xcorr_forward(x, w):
  """`layers.correlation` forward (layers.py:21-38): x [B,C,H,W], w [B,C,kh,kw] -> float32 [B,1,OH,OW].
  float32 features take the fp32 vector kernel (the reference's precision); bfloat16 features of the shapes in
  MFMA_SHAPES (the autocast rollout path) take the MFMA kernel."""
  if not x.is_cuda:
    raise RuntimeError('xcorr_forward needs a HIP device (no CPU fallback)')
  if x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.shape[-1] == x.shape[-2] and \
     w.shape[-1] == w.shape[-2] and (x.shape[-1], w.shape[-1]) in MFMA_SHAPES:
    return xcorr_forward_bf16(x, w)
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
