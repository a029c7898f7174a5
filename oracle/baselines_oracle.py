"""CPU restatement (vectorised numpy, float64) of the heuristic baseline policies.  TEST INFRASTRUCTURE ONLY.

Follows stackrl/baselines.py: `get_inputs` :21-26, `height` :28-43, `difference` :45-77, `corrcoef` :79-114,
`correlate` :141-143, `goal_overlap` :152-156, `Baseline.call` :201-217.
PINNED: checked against golden vectors produced by the reference's own baselines.py
(tests/golden/make_baselines_golden.py -> baselines_golden.npz)."""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view


def get_inputs(inputs):
  gmax = inputs[0][:, :, 1].max()                      # baselines.py:23
  return inputs[0][:, :, 0] / gmax, inputs[1][:, :, 0] / gmax


def _windows(o, n):
  return sliding_window_view(o, n.shape)               # [OH, OW, h, w]


def height(inputs):
  o, n = get_inputs(inputs)
  nw = n > 0
  return np.where(nw, _windows(o, n) + n, 0).max(axis=(2, 3))


def difference(inputs, difference_exponent=2, weights_exponent=2):
  o, n = get_inputs(inputs)
  nw = n > 0
  if weights_exponent > 0:
    wi = (np.arange(n.shape[0], dtype='float') - n.shape[0] / 2) ** 2
    wj = (np.arange(n.shape[1], dtype='float') - n.shape[1] / 2) ** 2
    w = (wi[:, None] + wj[None, :]) ** (weights_exponent / 2)
    w = np.where(nw, w, 0)
  else:
    w = nw.astype('float')
  w = w / w.sum()
  h = _windows(o, n) + n
  h0 = np.where(nw, h, 0).max(axis=(2, 3))
  return (w * np.abs(h0[:, :, None, None] - h) ** difference_exponent).sum(axis=(2, 3))


def corrcoef(inputs, localized=False):
  o, n = get_inputs(inputs)
  nw = (n > 0) if localized else np.ones_like(n, dtype=bool)
  cnt = np.count_nonzero(nw)
  n = n - np.sum(np.where(nw, n, 0)) / cnt
  n_var = np.sum(np.where(nw, n ** 2, 0))
  W = _windows(o, n)
  f = np.zeros(W.shape[:2])
  if n_var == 0:
    return f
  o_ = W - (np.where(nw, W, 0).sum(axis=(2, 3)) / cnt)[:, :, None, None]
  o_var = np.where(nw, o_ ** 2, 0).sum(axis=(2, 3))
  num = np.where(nw, n * o_, 0).sum(axis=(2, 3))
  ok = o_var != 0
  f[ok] = num[ok] / np.sqrt(n_var * o_var[ok])
  return f


def correlate(inputs):
  o, n = get_inputs(inputs)
  return (_windows(o, n) * n).sum(axis=(2, 3)) / n.sum()


def goal_overlap(inputs, threshold=0.75):
  b = (inputs[0][:, :, 0] < inputs[0][:, :, 1]).astype('int')
  n = (inputs[1][:, :, 0] > 0).astype('int')
  f = (_windows(b, n) * n).sum(axis=(2, 3))
  return f >= threshold * f.max()


METHODS = {'height': height, 'difference': difference, 'corrcoef': corrcoef, 'correlate': correlate}


def _minimum_filter_const0(v, size):
  """scipy.ndimage.minimum_filter(values, size, mode='constant') (cval = 0 outside the array)."""
  r = size // 2
  p = np.pad(v, r, mode='constant', constant_values=0.0)
  return sliding_window_view(p, (size, size)).min(axis=(2, 3))


def select(values, mask, goal=True, minorder=1):
  """Baseline.call (baselines.py:201-217): returns (flat action, negated value map)."""
  if goal:
    if minorder:
      minima = np.logical_and(mask, _minimum_filter_const0(values, 1 + 2 * minorder) == values)
      if np.any(minima):
        return int(np.argmin(np.where(minima, values, np.inf))), -np.where(mask, values, values[mask].max() + 0.001)
    return int(np.argmin(np.where(mask, values, np.inf))), -np.where(mask, values, values[mask].max() + 0.001)
  return int(np.argmin(values)), -values
