"""Greedy policies of the reference (stackrl/agents/policies.py): `Greedy` (:4-37) on a torch value estimator and the
batch-wise forms used with Stack-v2 (`TestStackEnv`), where one observation holds the 2^k orientations of the pending
rock and the action is `(orientation index, pixel)`."""
import torch


class Greedy(object):
  """`Greedy` (policies.py:4-37): arg-max of `model(inputs)` over the last axis; `batchwise=True` additionally picks
  the row with the largest maximum and returns `(row, action of that row)`; `value=True` also returns the values.
  Ties go to the lowest index in both steps (tf.math.argmax)."""

  def __init__(self, model, value=False, batchwise=False):
    if not callable(model):
      raise TypeError('model must be callable.')
    self.model, self.value, self.batchwise = model, value, batchwise

  @torch.no_grad()
  def __call__(self, inputs):
    values = self.model(inputs)
    outputs = torch.argmax(values, dim=-1)
    if self.batchwise:
      row = torch.argmax(values.amax(dim=-1))
      outputs = (row, outputs[row])
    return (outputs, values) if self.value else outputs


def expand_orientations(inputs):
  """Stack-v2 observation of the vectorised env — overhead maps [B,H,W,2] once, object maps [B,n,h,w,1] — to the
  reference's layout (env.py:472-480: the overhead map stacked n times): ([B*n,H,W,2], [B*n,h,w,1])."""
  m, o = inputs
  B, n = o.shape[0], o.shape[1]
  return m[:, None].expand(B, n, *m.shape[1:]).reshape(B * n, *m.shape[1:]), o.reshape(B * n, *o.shape[2:])


class OrientationGreedy(object):
  """Batch-wise greedy policy for the vectorised Stack-v2 env: for every env the values of all its orientations,
  `model` evaluated on the expanded observation, and the action `orientation * A + pixel` of the overall maximum —
  what `Greedy(batchwise=True)` returns as `(index, action)` for one env (arg-max per row, then the best row; ties to
  the lowest index, which is the lowest flat index).  `minimize=True` serves cost-like models (the heuristic
  baselines, baselines.py:201-217)."""

  def __init__(self, model, value=False, minimize=False):
    if not callable(model):
      raise TypeError('model must be callable.')
    self.model, self.value, self.minimize = model, value, minimize

  @torch.no_grad()
  def __call__(self, inputs, n_valid=None):
    """`n_valid` (ordering freedom): only the first `n_valid` object maps hold a rock (`env.num_maps_on_show`); the
    reference's observation simply has no more rows than that (env.py:596-608)."""
    B, n = inputs[1].shape[0], inputs[1].shape[1]
    values = self.model(expand_orientations(inputs)).reshape(B, n, -1)      # [B, n, A]
    if n_valid is not None and n_valid < n:
      values = values.clone()
      values[:, n_valid:] = float('inf') if self.minimize else -float('inf')
    values = values.reshape(B, -1)                                          # [B, n * A]
    actions = torch.argmin(values, dim=-1) if self.minimize else torch.argmax(values, dim=-1)
    return (actions, values) if self.value else actions
