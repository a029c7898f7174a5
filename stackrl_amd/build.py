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
# -fno-slp-vectorize: with the SLP vectoriser's packed-fp32 code (v_pk_fma / v_pk_mul / v_pk_add_f32, v_mov_b64) the settle
#   kernel returned, in a few envs per thousand, results that differ from the oracle's — but only while convolution kernels of
#   the Q-net ran on another stream; alone it was bit-exact.  Without it no difference was ever seen (DESIGN.md section 6a,
#   tests/diag/diag_conc.py, tests/test_parity_gpu.py::test_env_step_under_the_concurrent_rollout_forward...); speed is the same.
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
QFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-unused-function',
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
