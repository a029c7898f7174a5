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
# -fno-slp-vectorize: on gfx950 a packed-fp32 instruction whose LOW lane takes the HIGH half of its second source
#   (v_pk_add_f32 / v_pk_mul_f32 ... op_sel:[x,1]) reads 0 for that operand now and then while wavefronts of an MFMA kernel
#   share the CU (tools/experiments/pk_seq2.hip reproduces it in 30 lines).  clang's SLP vectoriser
#   emits exactly that form when it packs scalars that sit in different halves of their pairs; with it the settle and render
#   kernels returned results that differ from the oracle's in a few envs per thousand steps, only beside the Q-net's
#   convolution kernels.  Without the vectoriser neither library contains the form (tests/test_isa_guard.py checks the
#   ISA); the hand-written packed FMAs of the ray cast select halves of their FIRST source only, which is not affected.
#   Speed: settle -3.4 %, everything else unchanged.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-fno-slp-vectorize',
         '-fno-fast-math', '-Wall', '-Wno-unused-function', '-Wno-unused-value', '-Wno-unused-result']


def stale():
  if not os.path.isfile(LIB):
    return True
  t = os.path.getmtime(LIB)
  return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


QLIB = os.path.join(HERE, 'libstackrl_qnet.so')
QSRC = ['qnet.hip', 'heuristics.hip', 'xcorr_mfma.hip', 'epilogue.hip', 'conv_mfma.hip', 'conv_gemm.hip', 'learner.hip',
        'train_conv.hip']
QDEPS = QSRC + [ os.path.join('..', '..', 'include', 'stackrl_qnet.h')]
# the Q-net ops are ordinary fp32 kernels compared against a torch fp32 reference with a stated tolerance
# (-fno-slp-vectorize: see above; it costs the Q-net's kernels nothing measurable — 20.6 against 20.8 ms per 2,048-sample forward)
QFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fno-slp-vectorize', '-Wall', '-Wno-unused-function',
          '-Wno-unused-value', '-Wno-unused-result']


def qstale():
  if not os.path.isfile(QLIB):
    return True
  t = os.path.getmtime(QLIB)
  return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in QDEPS)


def build(force=False, verbose=False):
  hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
  if force or stale():
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB]
    if verbose:
      print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
  if force or qstale():
    cmd = [hipcc] + QFLAGS + [os.path.join(CSRC, f) for f in QSRC] + ['-o', QLIB]
    if verbose:
      print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
  return LIB


if __name__ == '__main__':
  build(force='-f' in sys.argv, verbose=True)
