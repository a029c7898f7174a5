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
    L.srl_policy_head.restype = ctypes.c_int
    L.srl_policy_head.argtypes = [VP, VP, VP, ctypes.c_float, VP, ctypes.c_int32, ctypes.c_int32, VP]
    L.srl_qnet_last_error.restype = ctypes.c_char_p
    _LIB = L
  return _LIB


def _stream(t):
  return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def xcorr_forward(x, w):
  """`layers.correlation` forward (layers.py:21-38): x [B,C,H,W], w [B,C,kh,kw] float32 -> [B,1,OH,OW]."""
  if not x.is_cuda:
    raise RuntimeError('xcorr_forward needs a HIP device (no CPU fallback)')
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
