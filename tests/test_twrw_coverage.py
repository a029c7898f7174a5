"""The split-K bookkeeping of the update's weight-gradient kernels, enumerated on the host (no GPU).

`k_twrw` (stackrl_amd/csrc/train_conv.hip:254-322) leaves ONE partial sum per workgroup in a scratch buffer
`partial[G][taps][cin16][cout]` and `k_twrw_finish` (:330-365) adds the G partials of every element in index order.  Round 3
recorded one weight-gradient mismatch of 9e-2 for the shape (256 -> 256, 4 x 8 x 8) that never came back; one candidate was a
slot the finishing kernel reads without any workgroup having written it.  This test restates the two kernels' index
arithmetic — grid, tile ranges per group, wave / k-step / lane -> pixel, lane -> (input channel, output channel) — and
checks, for that shape and for every layer shape of `DeepQSiamFCN` at the update's batch sizes, that

  * every element of the scratch the finish kernel reads is written by exactly one (workgroup, lane, register),
  * every pixel tile is reduced by exactly one group (groups past the last tile write zeros, not nothing),
  * every pixel of a tile is taken by exactly one (wave, k-step, lane row),
  * the scratch size the library reports (`srl_twrw_scratch_floats`, the C-ABI) is the size the kernels index.

On the GPU the same property is held dynamically: tests/test_train_conv_gpu.py hands the kernels a NaN-filled scratch."""
import ctypes

import numpy as np
import pytest


def wrw_tiles(taps, B, H, W):
  if taps == 1:
    return (B * H * W + 255) // 256
  if W > 8 or H > 8:
    return B * ((H + 15) // 16) * ((W + 15) // 16)
  return ((B + 3) // 4) * ((H + 7) // 8) * ((W + 7) // 8)


def wrw_groups(taps, B, H, W, cin, cout):
  tiles = wrw_tiles(taps, B, H, W)
  cot = 4 if cout % 64 == 0 else 2 if cout % 32 == 0 else 1
  blocks = ((cin + 15) // 16) * (cout // (cot * 16))
  return int(min(max(1024 // blocks, 1), tiles, 512)), cot


def net_shapes():
  """(taps, H, cin, cout) of every convolution `HandNet` differentiates (nets.py / layers.py:135-259) at 128 / 32 inputs."""
  out = []
  for res, cin0, depth in ((128, 2, 4), (32, 1, 2)):
    c, r = cin0, res
    for i in range(depth):
      f = 16 * 2 ** i
      out += [(9, r, c, f), (9, r, f, f)]
      c, r = f, r // 2
    fb = 16 * 2 ** depth
    out += [(9, r, c, fb), (9, r, fb, fb)]
    c = fb
    for i in range(depth - 1, -1, -1):
      f = 16 * 2 ** i
      out.append((1, r, c, 4 * f))          # up{i}: 1 x 1 to 4 f channels (depth-to-space)
      r *= 2
      out += [(9, r, 2 * f, f), (9, r, f, f)]
      c = f
  out += [(9, 97, 1, 16), (9, 97, 16, 16)]     # pos_layers
  return out


SHAPES = [(9, 4, 8, 8, 256, 256)] + [(t, B, r, r, ci, co) for (t, r, ci, co) in net_shapes() for B in (32, 3)]


@pytest.mark.parametrize('taps,B,H,W,cin,cout', sorted(set(SHAPES)))
def test_every_partial_the_finish_kernel_reads_is_written_exactly_once(taps, B, H, W, cin, cout):
  G, cot = wrw_groups(taps, B, H, W, cin, cout)
  cin16 = (cin + 15) // 16 * 16
  n = taps * cin16 * cout
  writes = np.zeros(G * n, np.int32)
  lane = np.arange(64)
  nci, ncz = (cin + 15) // 16, cout // (cot * 16)
  for g in range(G):
    for by in range(nci):
      for bz in range(ncz):
        ci0, co0 = by * 16, bz * cot * 16
        for t in range(taps):
          for ct in range(cot):
            co = co0 + ct * 16 + (lane & 15)
            ok = co < cout
            for i in range(4):
              idx = g * n + (t * cin16 + ci0 + 4 * (lane >> 4) + i) * cout + co
              np.add.at(writes, idx[ok], 1)
  assert writes.min() == 1 and writes.max() == 1, 'slots written {} .. {} times'.format(writes.min(), writes.max())
  # the finish kernel: block x, thread t -> element e = 32 x + (t & 31), partials g = (t >> 5), + 8, ... < G: every (g, e) once
  reads = np.zeros(G * n, np.int32)
  nwb = (n + 31) // 32
  e = (np.arange(nwb)[:, None] * 32 + np.arange(32)[None, :]).reshape(-1)
  e = e[e < n]
  for r in range(8):
    for g in range(r, G, 8):
      reads[g * n + e] += 1
  assert reads.min() == 1 and reads.max() == 1


@pytest.mark.parametrize('taps,B,H,W,cin,cout', sorted(set(SHAPES)))
def test_every_pixel_is_reduced_exactly_once(taps, B, H, W, cin, cout):
  G, _ = wrw_groups(taps, B, H, W, cin, cout)
  tiles = wrw_tiles(taps, B, H, W)
  per = (tiles + G - 1) // G
  seen_tile = np.zeros(tiles, np.int32)
  for g in range(G):
    t0, t1 = g * per, min(g * per + per, tiles)
    if t0 < t1:
      seen_tile[t0:t1] += 1
  assert (seen_tile == 1).all()
  # pixel q = 64 wave + 4 k-step + lane row of a tile -> (sample, y, x), bounds-checked like the kernel
  seen = np.zeros((B, H, W), np.int32)
  q = (64 * np.arange(4)[:, None, None] + 4 * np.arange(16)[None, :, None] + np.arange(4)[None, None, :]).reshape(-1)
  assert sorted(q.tolist()) == list(range(256))
  for tile in range(tiles):
    if taps == 1:
      p = tile * 256 + q
      p = p[p < B * H * W]
      np.add.at(seen.reshape(-1), p, 1)
      continue
    TW = 16 if (W > 8 or H > 8) else 8
    NS = 1 if TW == 16 else 4
    tx, ty = (W + TW - 1) // TW, (H + TW - 1) // TW
    gidx, r = divmod(tile, tx * ty)
    b0, y0, x0 = gidx * NS, (r // tx) * TW, (r % tx) * TW
    if TW == 16:
      s, yy, xx = np.zeros_like(q), q >> 4, q & 15
    else:
      s, yy, xx = q >> 6, (q >> 3) & 7, q & 7
    pb, py, px = b0 + s, y0 + yy, x0 + xx
    ok = (pb < B) & (py < H) & (px < W)
    np.add.at(seen, (pb[ok], py[ok], px[ok]), 1)
  assert seen.min() == 1 and seen.max() == 1


def test_the_library_reports_the_scratch_size_the_kernels_index():
  from stackrl_amd import build
  build.build()
  L = ctypes.CDLL(build.QLIB)       # host-side arithmetic only: no GPU call
  L.srl_twrw_scratch_floats.restype = ctypes.c_int64
  L.srl_twrw_scratch_floats.argtypes = [ctypes.c_int32] * 6
  for taps, B, H, W, cin, cout in sorted(set(SHAPES)):
    G, _ = wrw_groups(taps, B, H, W, cin, cout)
    assert L.srl_twrw_scratch_floats(B, H, W, cin, cout, taps) == G * taps * ((cin + 15) // 16 * 16) * cout, (taps, B, H, W, cin, cout)
