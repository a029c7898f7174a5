"""Weights of the reference's Keras `DeepQSiamFCN` (models.py:106-201) <-> the torch net (stackrl_amd/nets.py).

The reference saves `net.save_weights(.../saved_weights/<iter>/weights)` (training.py:495-505), a TensorFlow checkpoint
that only TensorFlow reads.  Someone who has TensorFlow exports it once,

    np.savez('weights.npz', **{w.name: w.numpy() for w in net.weights})

and `load_keras_weights(net, 'weights.npz')` maps it onto the torch net by the reference's layer names:
`Left/convdw{i}{j}`, `Left/conv{depth}{j}`, `Left/up{i}`, `Left/convuw{i}{j}` and the same under `Right/`
(layers.py:204-253); the unnamed layers get Keras' automatic names, taken here in creation order — `dense*` = the
dueling stream (layers.py:424-436), `conv2d*` = the layers after the correlation (layers.py:439-472).

Layouts: Conv2D kernel HWIO -> OIHW; Conv2DTranspose kernel (kh, kw, out, in) -> (in, out, kh, kw); Dense kernel
(in, out) -> (out, in).  Both frameworks correlate (no kernel flip) and `Concatenate([up, skip])` matches
`torch.cat([x, skip], 1)`, so no channel permutation is needed."""
import re

import numpy as np
import torch


def _strip(name):
  return re.sub(r':\d+$', '', name)


def _auto_order(names, stem):
  """Keras' automatic names `stem`, `stem_1`, ... (possibly starting later, if other models were built first) in
  creation order."""
  found = {}
  for n in names:
    m = re.fullmatch(r'(?:.*/)?{}(?:_(\d+))?'.format(stem), n)
    if m:
      found[int(m.group(1) or 0)] = n
  return [found[k] for k in sorted(found)]


def _unet_names(prefix, unet):
  """(keras layer name, torch module, kind) for one U-Net, in the reference's creation order."""
  out = []
  d = unet.depth
  for i in range(d):
    out += [('{}/convdw{}{}'.format(prefix, i, j), unet.down[i][2 * j], 'conv') for j in range(2)]
  out += [('{}/conv{}{}'.format(prefix, d, j), unet.bottom[2 * j], 'conv') for j in range(2)]
  for n, i in enumerate(range(d - 1, -1, -1)):
    out.append(('{}/up{}'.format(prefix, i), unet.up[n], 'convt'))
    out += [('{}/convuw{}{}'.format(prefix, i, j), unet.upconv[n][2 * j], 'conv') for j in range(2)]
  return out


def layer_table(net, names=None):
  """[(keras layer name, torch module, kind)] for the whole net.  `names`: the layer names present in a checkpoint, to
  resolve the automatic names; without it the first-model names (`dense`, `dense_1`, `conv2d`, ...) are used."""
  table = _unet_names('Left', net.left) + _unet_names('Right', net.right)
  dense = [m for m in getattr(net, 'value', []) if isinstance(m, torch.nn.Linear)]
  pos = [m for m in net.pos if isinstance(m, torch.nn.Conv2d)]
  def auto(stem, n):
    if names is not None:
      got = _auto_order(names, stem)
      if len(got) != n:
        raise ValueError('expected {} layers named {}*, found {}'.format(n, stem, got))
      return got
    return [stem if k == 0 else '{}_{}'.format(stem, k) for k in range(n)]
  table += [(n, m, 'dense') for n, m in zip(auto('dense', len(dense)), dense)]
  table += [(n, m, 'conv') for n, m in zip(auto('conv2d', len(pos)), pos)]
  return table


def _to_torch(kind, kernel):
  if kind == 'dense':
    return np.ascontiguousarray(kernel.T)
  return np.ascontiguousarray(kernel.transpose(3, 2, 0, 1))      # HWIO -> OIHW;  (kh, kw, out, in) -> (in, out, kh, kw)


def _to_keras(kind, weight):
  if kind == 'dense':
    return np.ascontiguousarray(weight.T)
  return np.ascontiguousarray(weight.transpose(2, 3, 1, 0))


def load_keras_weights(net, weights):
  """`weights`: a path to the `.npz` described above or a mapping `variable name -> array` (`.../kernel:0`,
  `.../bias:0`).  Every torch parameter must be matched and every shape must agree."""
  if isinstance(weights, str):
    with np.load(weights) as z:
      weights = {k: z[k] for k in z.files}
  weights = {_strip(k): np.asarray(v) for k, v in weights.items()}
  layers = sorted({k.rsplit('/', 1)[0] for k in weights})
  used = set()
  with torch.no_grad():
    for name, mod, kind in layer_table(net, layers):
      for var, par in (('kernel', mod.weight), ('bias', mod.bias)):
        key = '{}/{}'.format(name, var)
        if key not in weights:
          raise KeyError('variable {} is missing from the checkpoint'.format(key))
        a = weights[key] if var == 'bias' else _to_torch(kind, weights[key])
        if tuple(a.shape) != tuple(par.shape):
          raise ValueError('{}: shape {} does not fit {}'.format(key, a.shape, tuple(par.shape)))
        par.copy_(torch.from_numpy(a).to(par.dtype))
        used.add(key)
  extra = sorted(set(weights) - used)
  if extra:
    raise ValueError('variables of the checkpoint left unused: {}'.format(extra))
  return net


def export_keras_weights(net):
  """The inverse: `{'<layer>/kernel:0': array, '<layer>/bias:0': array}` in the reference's naming and layouts."""
  out = {}
  for name, mod, kind in layer_table(net):
    out['{}/kernel:0'.format(name)] = _to_keras(kind, mod.weight.detach().cpu().numpy())
    out['{}/bias:0'.format(name)] = mod.bias.detach().cpu().numpy().copy()
  return out
