"""Work-around for the packed-fp32 instruction form that misreads an operand on gfx950 beside certain MFMA wavefronts
(DESIGN.md section 6a).

A `v_pk_add_f32` / `v_pk_mul_f32` / `v_pk_fma_f32` whose LOW lane takes the HIGH half of its SECOND source (`op_sel:[x,1..]`) —
or, for the fma, of its addend (`op_sel:[x,x,1]`) — reads 0 for that operand now and then while another wavefront of the CU
issues an MFMA with 128-bit A / B operands (gfx950's `v_mfma_f32_16x16x32_bf16` / `_f16`, `v_mfma_f32_32x32x16_bf16`,
`v_mfma_i32_16x16x64_i8`; not `v_mfma_f32_16x16x4_f32`, not the 64-bit-operand `v_mfma_f32_16x16x16_bf16`, not vector-ALU work
of any kind: tools/experiments/pk_seq2.hip beside tools/experiments/pk_aggressor.hip).  The same selection on the FIRST source is
clean (same experiment), and the first two sources of all three instructions commute, so the second-source form is removed
from compiled assembly by swapping the two sources together with their modifier bits — the instruction computes the same
lanes from the same registers.  (An addend with the selection cannot be repaired that way: `rewrite` reports it and build.py
falls back to the build without the vectoriser; the compiler of this image emits none.)  `rewrite` does that on the text of a
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
    return line, None                      # the addend selects its high half for the low lane: no commuting partner
  if sel[1] != 1:
    return line, False
  if sel[0] == 1:
    return line, None                      # both sources select their high half for the low lane: a swap does not help
  if any(len(v) != nsrc for v in mods.values()):
    return line, None
  operands[1], operands[2] = operands[2], operands[1]
  for v in mods.values():
    v[0], v[1] = v[1], v[0]
  default = {'op_sel': [0] * nsrc, 'op_sel_hi': [1] * nsrc, 'neg_lo': [0] * nsrc, 'neg_hi': [0] * nsrc}
  tail = ' '.join('%s:[%s]' % (k, ','.join(map(str, mods[k]))) for k in _ORDER if k in mods and mods[k] != default[k])
  return '%s%s %s%s%s' % (indent, op, ', '.join(operands), (' ' + tail) if tail else '', comment or ''), True


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
