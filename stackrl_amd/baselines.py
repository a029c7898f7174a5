"""Heuristic baseline policies (`stackrl/baselines.py`) on the device: the reference's quality yardstick and its
optional initial-collect policy (`training.py:256-263`, `config.gin:118`, `:142`).

`Baseline(method=..., goal=True, minorder=1)` mirrors `stackrl.Baseline` (baselines.py:168-217): called with a batch of
observations it returns one flat action per env (and the negated value maps with `value=True`).  The sliding-window
value maps and the selection run in libstackrl_qnet.so (csrc/heuristics.hip); there is no CPU fallback.
`random` (baselines.py:145-150) draws uniform values with torch's generator (numpy's default_rng stream is not reproduced).
"""
import ctypes

import torch

from stackrl_amd import qops

METHODS = {'random': 0, 'correlate': 1, 'height': 2, 'difference': 3, 'corrcoef': 4}   # baselines.py:158-165


def _lib():
  L = qops.load()
  if not getattr(L, '_heur_ready', False):
    VP = ctypes.c_void_p
    L.srl_heuristic.restype = ctypes.c_int
    L.srl_heuristic.argtypes = [ctypes.c_int32, VP, VP, VP, VP] + [ctypes.c_int32] * 6 + [ctypes.c_double, VP]
    L.srl_baseline_select.restype = ctypes.c_int
    L.srl_baseline_select.argtypes = [VP, VP, ctypes.c_int32, ctypes.c_int32, VP, VP, ctypes.c_int32, ctypes.c_int32, VP]
    L._heur_ready = True
  return L


def _stream(t):
  return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def heuristic_values(method, inputs, mask=True, difference_exponent=2, weights_exponent=2, localized=False,
                     threshold=0.75, generator=None):
  """Value map float64 [B, OH, OW] of one method and (optionally) the goal-overlap mask bool [B, OH, OW]."""
  xm, xo = inputs
  if not xm.is_cuda:
    raise RuntimeError('heuristic_values needs a HIP device (no CPU fallback)')
  xm = xm.contiguous(); xo = xo.contiguous()
  B, H = xm.shape[0], xm.shape[1]
  h = xo.shape[1]
  OH = H - h + 1
  mid = METHODS[method] if isinstance(method, str) else int(method)
  vals = torch.empty((B, OH, OH), dtype=torch.float64, device=xm.device)
  mk = torch.empty((B, OH, OH), dtype=torch.uint8, device=xm.device) if mask else None
  L = _lib()
  with torch.cuda.device(xm.device):
    rc = L.srl_heuristic(mid if mid else 2, xm.data_ptr(), xo.data_ptr(), vals.data_ptr(), mk.data_ptr() if mask else None,
                         B, H, h, int(difference_exponent), int(weights_exponent), int(bool(localized)), float(threshold),
                         _stream(xm))
  if rc:
    raise RuntimeError(L.srl_qnet_last_error().decode())
  if mid == 0:   # 'random' keeps the mask of the heuristic pass and replaces the values
    vals = torch.rand(vals.shape, generator=generator, device=xm.device, dtype=torch.float64)
  return (vals, mk.bool()) if mask else vals


def select(values, mask=None, goal=True, minorder=1, value=False):
  """`Baseline.call` (baselines.py:201-217) on value maps float64 [B, OH, OW] (+ mask) -> actions int64 [B]."""
  values = values.contiguous()
  B, OH = values.shape[0], values.shape[1]
  mk = mask.to(torch.uint8).contiguous() if (goal and mask is not None) else None
  if goal and mk is None:
    raise ValueError('goal=True needs the goal-overlap mask')
  actions = torch.empty(B, dtype=torch.int64, device=values.device)
  neg = torch.empty_like(values) if value else None
  L = _lib()
  with torch.cuda.device(values.device):
    rc = L.srl_baseline_select(values.data_ptr(), mk.data_ptr() if mk is not None else None, int(bool(goal)), int(minorder),
                               actions.data_ptr(), neg.data_ptr() if value else None, B, OH, _stream(values))
  if rc:
    raise RuntimeError(L.srl_qnet_last_error().decode())
  return (actions, neg) if value else actions


class Baseline(object):
  """stackrl.Baseline (baselines.py:168-217)."""

  def __init__(self, method='random', goal=True, minorder=1, value=False, seed=None, **kwargs):
    if isinstance(method, str):
      if method not in METHODS:
        raise ValueError('Invalid value {} for argument method. Must be in {}'.format(method, list(METHODS)))   # :184-187
    else:
      raise TypeError('Invalid type {} for argument method.'.format(type(method)))
    self.method, self.goal, self.minorder, self.value, self.kwargs = method, goal, minorder, value, kwargs
    self._seed, self._gen = seed, None

  def __call__(self, inputs):
    if self.method == 'random' and self._gen is None:
      self._gen = torch.Generator(device=inputs[0].device)
      if self._seed is not None:
        self._gen.manual_seed(int(self._seed))
    kw = {k: v for k, v in self.kwargs.items() if k in ('difference_exponent', 'weights_exponent', 'localized', 'threshold')}
    out = heuristic_values(self.method, inputs, mask=self.goal, generator=self._gen, **kw)
    vals, mk = out if self.goal else (out, None)
    return select(vals, mk, goal=self.goal, minorder=self.minorder, value=self.value)
