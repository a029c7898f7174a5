"""Index arithmetic of csrc/conv_gemm.hip restated in Python (CPU): two claims the kernel's comments make and its speed rests on.

1. `GemmCfg::tile_delta(t)`: the LDS offset of a lane's pixel in N tile t is its offset in tile 0 plus a constant that does not
   depend on the lane — for every layer shape the dispatcher instantiates (`dispatch()`), so the eight offsets are one base
   register plus immediates.
2. SWZ (fp32-class, 64 output channels at 32 x 32): with the 64-byte pixel stride and the chunk slot `chunk ^ ((column >> 2) & 3)`
   the sixteen lanes of one 16-byte fragment read land in sixteen different bank quads, for every tap column offset and every
   k-group; and the write side (staging) and the read side (`sw[dx]`) name the same slot.
"""
import pytest

# (cin is irrelevant to the layout) (COUT, W) of every SRL_CASE in dispatch(), 128^2 and 64^2 observations
SHAPES = [(64, 32), (128, 16), (256, 8), (64, 8), (64, 16), (128, 8), (256, 4)]


def cfg(cout, w, x3):
  wm = cout // 64
  wn = 4 // wm
  pxt = 128 * wn
  ni = pxt // (w * w) if pxt >= w * w else 1
  rt = w if ni > 1 else pxt // w
  th, twd = rt + 2, w + 2
  swz = x3 and cout == 64 and w == 32
  return dict(WN=wn, PXT=pxt, NI=ni, RT=rt, TH=th, TWD=twd, SWZ=swz, PS=32 if swz else 40)


def tile_delta(c, w, t):
  q = 16 * t
  return ((q // (c['RT'] * w) * c['TH'] + (q // w) % c['RT']) * c['TWD'] + q % w) * c['PS']


@pytest.mark.parametrize('cout,w', SHAPES)
@pytest.mark.parametrize('x3', [False, True])
def test_tile_offsets_are_base_plus_a_lane_independent_constant(cout, w, x3):
  c = cfg(cout, w, x3)
  for wn in range(c['WN']):
    for n in range(16):
      def off(t):
        p = wn * 128 + t * 16 + n                       # pixel within the workgroup tile, as the kernel's poff0 states it
        im, r, col = p // (c['RT'] * w), (p // w) % c['RT'], p % w
        return ((im * c['TH'] + r) * c['TWD'] + col) * c['PS']
      for t in range(8):
        assert off(t) - off(0) == tile_delta(c, w, t), (cout, w, x3, wn, n, t)
  # the immediates (tile + tap offset, in bytes) fit the 16-bit offset field of the LDS read
  worst = 2 * (tile_delta(c, w, 7) + (2 * c['TWD'] + 2) * c['PS'])
  assert worst < 65536


def test_swizzled_fragment_reads_are_conflict_free_and_match_the_staging():
  c = cfg(64, 32, True)
  assert c['SWZ'] and c['PS'] == 32
  for t in range(8):
    for dx in range(3):
      for g in range(4):
        quads = set()
        for n in range(16):
          col = n + 16 * (t & 1) + dx                                   # tile column of the lane's pixel for this tap
          key_read = ((n + dx) >> 2) & 3                                # what the kernel keeps per lane: depends on n + dx only
          assert key_read == (col >> 2) & 3
          slot = g ^ key_read                                           # read side: sw[dx] = 8 (g ^ key)
          assert slot == g ^ ((col >> 2) & 3)                           # staging side: slot = ch ^ ((cc >> 2) & 3) for chunk ch = g
          pixel = 3 * c['TWD'] + col                                    # any row: the row adds a multiple of 34 pixels
          byte = pixel * 64 + slot * 16
          quads.add((byte // 16) % 16)                                  # 64 banks of 4 bytes = 16 quads of 16 bytes
        assert len(quads) == 16, (t, dx, g, sorted(quads))
