"""Work-around for the packed-fp32 instruction form that misreads an operand on gfx950 beside certain MFMA wavefronts
(DESIGN.md section 6a).

A `v_pk_add_f32` / `v_pk_mul_f32` / `v_pk_fma_f32` whose LOW lane takes the HIGH half of its SECOND source (`op_sel:[x,1..]`) —
or, for the fma, of its addend (`op_sel:[x,x,1]`) — reads 0 for that operand now and then while another wavefront of the CU
issues an MFMA with 128-bit A / B operands (gfx950's `v_mfma_f32_16x16x32_bf16` / `_f16`, `v_mfma_f32_32x32x16_bf16`,
`v_mfma_i32_16x16x64_i8`; not `v_mfma_f32_16x16x4_f32`, not the 64-bit-operand `v_mfma_f32_16x16x16_bf16`, not vector-ALU work
of any kind: tools/experiments/pk_seq2.hip beside tools/experiments/pk_aggressor.hip).  The same selection on the FIRST source is
clean (same experiment), and the first two sources of all three instructions commute, so the second-source form is removed
from compiled assembly by swapping the two sources together with their modifier bits — the instruction computes the same
lanes from the same registers.  (An addend with the selection, or both sources with it, cannot be repaired that way: such an instruction
is written as the two single-lane instructions it stands for, `_split`; what that cannot express is reported and build.py
falls back to the build without the vectoriser.)  `rewrite` does that on the text of a
gfx950 assembly file; `flagged` lists what is (still) there — in compiler output or in the disassembly of a built library
(`shipped_asm`).  Used by build.py on the env library (which is compiled with clang's SLP vectoriser, the producer of the
form), by tests/test_isa_guard.py and by tools/scan_third_party.py."""
import re

_INSN = re.compile(r'^(\s*)(v_pk_(?:add|mul|fma)_f32)\s+([^;]*?)(\s*;.*)?$')
_MOD = re.compile(r'\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]')
# second entry of op_sel = 1 (the second source's high half for the LOW lane), or the third (the fma's addend): both fail
BAD = re.compile(r'^\s*(v_pk_(?:add|mul|fma)_f32)\b.*\bop_sel:\[(?:[01],1|[01],[01],1)')
_ORDER = ('op_sel', 'op_sel_hi', 'neg_lo', 'neg_hi')


_OBJDUMP_LABEL = re.compile(r'^[0-9a-f]+ <([^>]+)>:$')


def flagged(text):
  """[(line number, kernel label, instruction)] of the instructions of the failing form in an assembly text (compiler
  output or `llvm-objdump -d` of a code object)."""
  out, label = [], None
  for i, line in enumerate(text.splitlines(), 1):
    if line and not line[0].isspace() and line.endswith(':') and not line.startswith('.'):
      m = _OBJDUMP_LABEL.match(line)
      label = m.group(1) if m else line[:-1]
    if BAD.match(line):
      out.append((i, label, line.strip()))
  return out


def _fix(line):
  m = _INSN.match(line)
  if not m:
    return line, False
  indent, op, rest, comment = m.groups()
  mm = _MOD.search(rest)
  operands = [o.strip() for o in (rest[:mm.start()] if mm else rest).strip().rstrip(',').split(',')]
  nsrc = len(operands) - 1
  mods = {k: [int(x) for x in v.split(',')] for k, v in _MOD.findall(rest)}
  sel = mods.get('op_sel')
  if nsrc < 2 or sel is None or len(sel) < 2:
    return line, False
  if len(sel) > 2 and sel[2] == 1:
    return _split(indent, op, operands, mods, comment, line)   # the addend selects its high half for the low lane: no commuting partner
  if sel[1] != 1:
    return line, False
  if sel[0] == 1:
    return _split(indent, op, operands, mods, comment, line)   # both sources select their high half for the low lane: a swap does not help
  if any(len(v) != nsrc for v in mods.values()):
    return line, None
  operands[1], operands[2] = operands[2], operands[1]
  for v in mods.values():
    v[0], v[1] = v[1], v[0]
  default = {'op_sel': [0] * nsrc, 'op_sel_hi': [1] * nsrc, 'neg_lo': [0] * nsrc, 'neg_hi': [0] * nsrc}
  tail = ' '.join('%s:[%s]' % (k, ','.join(map(str, mods[k]))) for k in _ORDER if k in mods and mods[k] != default[k])
  return '%s%s %s%s%s' % (indent, op, ', '.join(operands), (' ' + tail) if tail else '', comment or ''), True


_VPAIR = re.compile(r'^v\[(\d+):(\d+)\]$')
_SCALAR = {'v_pk_mul_f32': 'v_mul_f32_e64', 'v_pk_add_f32': 'v_add_f32_e64', 'v_pk_fma_f32': 'v_fma_f32'}


def _split(indent, op, operands, mods, comment, line):
  """A flagged instruction that a swap of its sources cannot repair (both sources, or the fma's addend, select the high half
  for the low lane), written as the two single-lane instructions it stands for — `v_mul_f32` / `v_add_f32` / `v_fma_f32` are
  the same IEEE operations per lane — in an order in which neither overwrites a register the other still reads; when each
  lane's destination is a source of the other lane (`v_pk_mul_f32 v[a:b], v[a:b], v[c:d] op_sel:[1,1] op_sel_hi:[0,1]`, the one
  shape the compiler of this image emits) the two results are computed into the opposite halves and exchanged by
  `v_swap_b32`.  Only plain VGPR-pair operands with op_sel / op_sel_hi / neg_lo / neg_hi are handled; anything else is
  reported as not rewritable (None) and build.py falls back to the build without the vectoriser."""
  nsrc = len(operands) - 1
  regs = []
  for o in operands:
    m = _VPAIR.match(o)
    if not m or int(m.group(2)) != int(m.group(1)) + 1:
      return line, None
    regs.append(int(m.group(1)))
  if any(len(v) != nsrc for v in mods.values()) or any(k not in _ORDER for k in mods):
    return line, None
  sel = mods.get('op_sel', [0] * nsrc)
  sel_hi = mods.get('op_sel_hi', [1] * nsrc)
  neg_lo = mods.get('neg_lo', [0] * nsrc)
  neg_hi = mods.get('neg_hi', [0] * nsrc)
  d = regs[0]
  lo_src = [regs[1 + k] + sel[k] for k in range(nsrc)]         # registers the low lane reads
  hi_src = [regs[1 + k] + sel_hi[k] for k in range(nsrc)]      # registers the high lane reads

  def insn(dst, src, neg):
    return '%s%s v%d, %s' % (indent, _SCALAR[op], dst, ', '.join(('-' if n else '') + 'v%d' % r for r, n in zip(src, neg)))

  tag = (comment or '') + ' ; isa_fix: split of ' + ' '.join(line.split())
  if d not in hi_src:                       # the low lane's destination is not read by the high lane: low first
    return '\n'.join([insn(d, lo_src, neg_lo) + tag, insn(d + 1, hi_src, neg_hi)]), True
  if d + 1 not in lo_src:                   # the high lane's destination is not read by the low lane: high first
    return '\n'.join([insn(d + 1, hi_src, neg_hi) + tag, insn(d, lo_src, neg_lo)]), True
  # each destination is a source of the other lane: the high lane's result into the low register and vice versa (each then
  # overwrites only a register the other does not read — checked), and exchange
  if d + 1 not in hi_src and d not in lo_src:
    return '\n'.join([insn(d, hi_src, neg_hi) + tag, insn(d + 1, lo_src, neg_lo), '%sv_swap_b32 v%d, v%d' % (indent, d, d + 1)]), True
  return line, None


def rewrite(text):
  """(new text, number rewritten, number that could not be rewritten)."""
  out, n, left = [], 0, 0
  for line in text.split('\n'):
    new, done = _fix(line)
    if done:
      n += 1
    elif done is None:
      left += 1
    out.append(new)
  return '\n'.join(out), n, left


# ------------------------------------------------------------------------------------------------ shipped artefacts
_MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(path, arch='gfx950', llvm_bin='/opt/rocm/lib/llvm/bin'):
  """The device code objects (bytes) embedded in a built shared library: the `.hip_fatbin` section holds one clang offload
  bundle per translation unit, each a table of (offset, size, target triple) entries."""
  import struct
  import subprocess
  import tempfile
  with tempfile.TemporaryDirectory() as tmp:
    fat = tmp + '/fat.bin'
    subprocess.run([llvm_bin + '/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fat, path, tmp + '/stripped'], check=True)
    data = open(fat, 'rb').read()
  out, pos = [], data.find(_MAGIC)
  while pos >= 0:
    n, = struct.unpack_from('<Q', data, pos + len(_MAGIC))
    q = pos + len(_MAGIC) + 8
    for _ in range(n):
      off, size, tlen = struct.unpack_from('<QQQ', data, q)
      triple = data[q + 24:q + 24 + tlen].decode()
      q += 24 + tlen
      if triple.endswith(arch) and size:
        out.append(data[pos + off:pos + off + size])
    pos = data.find(_MAGIC, pos + len(_MAGIC))
  return out


def shipped_asm(path, arch='gfx950', llvm_bin='/opt/rocm/lib/llvm/bin'):
  """Disassembly (`llvm-objdump -d`) of every device code object of a built library, one text per translation unit."""
  import subprocess
  import tempfile
  texts = []
  for co in code_objects(path, arch, llvm_bin):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
      f.write(co)
      f.flush()
      texts.append(subprocess.run([llvm_bin + '/llvm-objdump', '-d', '--mcpu=' + arch, f.name], check=True,
                                  stdout=subprocess.PIPE, universal_newlines=True).stdout)
  return texts
