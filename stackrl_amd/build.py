"""Builds libstackrl_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the tree)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libstackrl_hip.so')
SOURCES = ['stackrl_hip.hip']
DEPS = ['stackrl_hip.hip', 'settle.hip', 'render.hip', 'srl_device.h', 'srl_kernels.h',
        os.path.join('..', '..', 'include', 'stackrl_hip.h'), os.path.join('..', '..', 'include', 'srl_types.h')]
# -ffp-contract=off: the solver/rasteriser definition is "one IEEE rounding per written operation"
# The packed-fp32 erratum: on the MI355X boxes of this pool a v_pk_add_f32 / v_pk_mul_f32 whose LOW lane
#   takes the HIGH half of its SECOND source (op_sel:[x,1]; v_pk_fma_f32 too, and its addend) reads 0 for that operand now and
#   then while another wavefront on the CU issues one of gfx950's MFMA shapes with 128-bit A / B operands
#   (v_mfma_f32_16x16x32_bf16 / _f16, v_mfma_f32_32x32x16_bf16, v_mfma_i32_16x16x64_i8): measured with a 30-line victim
#   (tools/experiments/pk_seq2.hip) beside one-property aggressors (tools/experiments/pk_aggressor.hip, tools/diag_aggressor.py,
#   profiles/r04_erratum_aggressor*.log) — 1 - 3 % of the executions beside a bare loop of such MFMAs, whether their
#   accumulators live in v or a registers, at any s_setprio, with or without wait states between them; 0 beside
#   v_mfma_f32_16x16x4_f32, the 64-bit-operand v_mfma_f32_16x16x16_bf16, vector FMAs, DPP, v_perm_b32, SDWA or packed FMAs, and
#   0 alone.  clang's SLP vectoriser emits exactly that form when it packs scalars that sit in different halves of their
#   pairs; with it the settle and render kernels returned results that differ from the oracle's in a few envs per thousand
#   steps, only beside the Q-net's bf16 convolution kernels.  The env library is therefore built in steps: device assembly
#   (vectoriser on: it is worth 3.4 % of the settle kernel), `isa_fix.rewrite` (the two commuting sources of every flagged
#   instruction swapped — the same selection on the FIRST source is clean), assembler, code object, fat binary, host object,
#   link: what hipcc does in one go, with the pass in the middle.  If any step fails the library is built in one go WITHOUT
#   the vectoriser instead (FLAGS_SAFE: no flagged instruction either, slower); the library says which it is
#   (`srl_build_info`, printed by bench.py).  tests/test_isa_guard.py checks the compiled ISA of every source file of both
#   libraries AND disassembles the shipped .so files.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
         '-fno-fast-math', '-Wall', '-Wno-unused-function', '-Wno-unused-value', '-Wno-unused-result']
FLAGS = FLAGS + os.environ.get('SRL_EXTRA_FLAGS', '').split()      # experiments only (e.g. -DSRL_STEP_PRIO=3)
FLAGS_SAFE = FLAGS + ['-fno-slp-vectorize']
LLVM_BIN = os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'lib', 'llvm', 'bin')
DEPS = DEPS + [os.path.join('..', 'isa_fix.py'), os.path.join('..', 'build.py')]


def source_hash(deps, flags):
  """sha256 over the contents of the files a library is built from and the flags it is built with (first 16 hex digits)."""
  import hashlib
  h = hashlib.sha256()
  for d in deps:
    with open(os.path.join(CSRC, d), 'rb') as f:
      h.update(d.encode() + b'\0' + f.read() + b'\0')
  h.update(' '.join(flags).encode())
  return h.hexdigest()[:16]


INFO_MARK = b'SRL_BUILD_INFO<'


def info(path):
  """What a built library says about itself — {'variant': ..., 'hash': ...} — read from the bytes of the file (the string
  `srl_build_info()` returns), or None for a file without it.  `stale()` compares the hash with the sources'."""
  try:
    with open(path, 'rb') as f:
      data = f.read()
  except OSError:
    return None
  i = data.find(INFO_MARK)
  if i < 0:
    return None
  j = data.find(b'>', i)
  variant, _, digest = data[i + len(INFO_MARK):j].decode().partition('|')
  return {'variant': variant, 'hash': digest}


def _info_flag(variant, digest):
  return '-DSRL_BUILD_INFO="{}{}|{}>"'.format(INFO_MARK.decode(), variant, digest)


VARIANT_FIXED = 'vectorised+rewritten'     # SLP vectoriser on, isa_fix.rewrite over the assembly
VARIANT_SAFE = 'safe'                      # one go without the SLP vectoriser (the fall-back; ~3 % slower settle kernel)


def device_asm(hipcc, flags, src):
  """gfx950 assembly of one source file, compiled with `flags` (no GPU needed)."""
  keep = [f for f in flags if f not in ('-shared', '-fPIC')]
  return subprocess.run([hipcc] + keep + ['--cuda-device-only', '-S', '-w', '-o', '-', src], check=True,
                        stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True).stdout


def fixed_env_asm(hipcc):
  """The env library's device assembly after the rewrite; raises if a flagged instruction is left."""
  from stackrl_amd import isa_fix
  text, n, left = isa_fix.rewrite(device_asm(hipcc, FLAGS, os.path.join(CSRC, SOURCES[0])))
  bad = isa_fix.flagged(text)
  if left or bad:
    raise RuntimeError('{} packed instructions of the failing form could not be rewritten: {}'.format(left or len(bad), bad[:3]))
  return text, n


def _build_env_fixed(hipcc, verbose):
  import tempfile
  src = os.path.join(CSRC, SOURCES[0])
  with tempfile.TemporaryDirectory() as tmp:
    def run(cmd):
      if verbose:
        print(' '.join(cmd), file=sys.stderr)
      subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL if not verbose else None)
    text, n = fixed_env_asm(hipcc)
    asm, obj, co, fb, host = (os.path.join(tmp, f) for f in ('dev.s', 'dev.o', 'dev.out', 'dev.hipfb', 'host.o'))
    with open(asm, 'w') as f:
      f.write(text)
    run([os.path.join(LLVM_BIN, 'clang'), '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', asm, '-o', obj])
    run([os.path.join(LLVM_BIN, 'lld'), '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', co, obj])
    run([os.path.join(LLVM_BIN, 'clang-offload-bundler'), '-type=o', '-bundle-align=4096',
         '-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950', '-input=/dev/null', '-input=' + co, '-output=' + fb])
    run([hipcc] + [f for f in FLAGS if f != '-shared'] + [_info_flag(VARIANT_FIXED, source_hash(DEPS, FLAGS)), '--cuda-host-only',
                                                         '-Xclang', '-fcuda-include-gpubinary', '-Xclang', fb, '-c', src, '-o', host])
    run([hipcc, '-shared', '-fPIC', host, '-o', LIB + '.tmp'])
    os.replace(LIB + '.tmp', LIB)
  if verbose:
    print('env library: {} packed instructions rewritten'.format(n), file=sys.stderr)


def stale():
  """The library is missing, or was not built from these sources with these flags (the hash it carries, not file times)."""
  i = info(LIB)
  return i is None or i['hash'] != source_hash(DEPS, FLAGS)


QLIB = os.path.join(HERE, 'libstackrl_qnet.so')
QSRC = ['qnet.hip', 'heuristics.hip', 'xcorr_mfma.hip', 'epilogue.hip', 'conv_mfma.hip', 'conv_gemm.hip', 'learner.hip',
        'train_conv.hip']
QDEPS = QSRC + ['srl_bf16.h', os.path.join('..', '..', 'include', 'stackrl_qnet.h')]
# the Q-net ops are ordinary fp32 kernels compared against a torch fp32 reference with a stated tolerance
# (-fno-slp-vectorize: see above; it costs the Q-net's kernels nothing measurable — 20.6 against 20.8 ms per 2,048-sample forward)
QFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fno-slp-vectorize', '-Wall', '-Wno-unused-function',
          '-Wno-unused-value', '-Wno-unused-result']


def qstale():
  i = info(QLIB)
  return i is None or i['hash'] != source_hash(QDEPS, QFLAGS)


def build(force=False, verbose=False):
  hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
  if force or stale():
    try:
      if os.environ.get('SRL_BUILD_SAFE'):
        raise RuntimeError('SRL_BUILD_SAFE is set')
      _build_env_fixed(hipcc, verbose)
    except Exception as e:       # any step of the long way round: the plain build without the vectoriser (same results, slower)
      print('stackrl_amd.build: env library built without the SLP vectoriser ({})'.format(str(e)[:200]), file=sys.stderr)
      cmd = [hipcc] + FLAGS_SAFE + [_info_flag(VARIANT_SAFE, source_hash(DEPS, FLAGS))] + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB]
      if verbose:
        print(' '.join(cmd), file=sys.stderr)
      subprocess.check_call(cmd)
  if force or qstale():
    cmd = [hipcc] + QFLAGS + [_info_flag('no-slp', source_hash(QDEPS, QFLAGS))] + [os.path.join(CSRC, f) for f in QSRC] + ['-o', QLIB]
    if verbose:
      print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
  return LIB


if __name__ == '__main__':
  build(force='-f' in sys.argv, verbose=True)
