"""CPU tests of the boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/stackrl_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  with open(os.path.join(ROOT, 'include', 'stackrl_hip.h')) as f:
    txt = f.read()
  return sorted(set(re.findall(r'\b(srl_[a-z0-9_]+)\s*\(', txt)))


def test_library_builds_and_exports_every_declared_symbol():
  from stackrl_amd import build, lib
  path = build.build()
  assert os.path.isfile(path)
  L = ctypes.CDLL(path)
  names = _declared()
  assert len(names) >= 18
  for n in names:
    assert hasattr(L, n), 'missing export ' + n
  assert sorted(lib.EXPORTS) == names, 'ctypes signature table out of sync with the header'


def test_qnet_library_exports_every_declared_symbol():
  from stackrl_amd import build
  build.build()
  with open(os.path.join(ROOT, 'include', 'stackrl_qnet.h')) as f:
    names = sorted(set(re.findall(r'\b(srl_[a-z0-9_]+)\s*\(', f.read())))
  assert names == ['srl_adam_step', 'srl_baseline_select', 'srl_bias_act', 'srl_bias_act_bwd_f32', 'srl_bias_act_bwd_scratch_floats', 'srl_bias_act_f32', 'srl_bias_act_pool', 'srl_bias_act_pool_f32', 'srl_conv3x3_bias_relu', 'srl_conv3x3_bias_relu_f32', 'srl_conv3x3_gemm_batch_multiple', 'srl_conv3x3_gemm_bias_relu', 'srl_conv3x3_gemm_supported', 'srl_conv3x3_gemm_wfrag_elems', 'srl_conv3x3_relu_project', 'srl_conv3x3_relu_project_f32', 'srl_conv3x3_thin', 'srl_conv3x3_thin_f32', 'srl_conv3x3_wfrag_elems', 'srl_conv_gemm_last_error',
                   'srl_conv_last_error', 'srl_convt2x2_bias_relu', 'srl_convt2x2_bias_relu_f32', 'srl_convt2x2_gemm_bias_relu', 'srl_convt2x2_gemm_supported', 'srl_convt2x2_wfrag_elems', 'srl_epilogue_last_error', 'srl_gumbel_topk',
                   'srl_gumbel_topk_scratch_bytes', 'srl_heuristic', 'srl_learner_last_error', 'srl_logit_extrema', 'srl_logit_extrema_scratch_bytes',
                   'srl_policy_head', 'srl_pool2x2', 'srl_qnet_build_info', 'srl_qnet_last_error', 'srl_replay_gather', 'srl_replay_scatter', 'srl_tact_bwd', 'srl_tact_bwd_blocks',
                   'srl_tact_bwd_scratch_floats', 'srl_tconv', 'srl_tcorr_grad', 'srl_td_epilogue', 'srl_tflip', 'srl_thead_bwd', 'srl_thead_fwd', 'srl_thin_conv3x3_bias_relu_f32', 'srl_thin_conv3x3_relu_project_f32', 'srl_tlayout', 'srl_train_conv_last_error', 'srl_trepack', 'srl_tu8_to_f32', 'srl_tvalue_bwd', 'srl_tvalue_fwd', 'srl_twrw', 'srl_twrw_scratch_floats', 'srl_xcorr_forward',
                   'srl_xcorr_mfma', 'srl_xcorr_mfma_last_error', 'srl_xcorr_mfma_scratch_bytes']
  L = ctypes.CDLL(build.QLIB)
  for n in names:
    assert hasattr(L, n), 'missing export ' + n


def test_config_struct_matches_header():
  from stackrl_amd.config import CConfig, StackConfig
  with open(os.path.join(ROOT, 'include', 'srl_types.h')) as f:
    txt = f.read()
  body = txt[txt.index('typedef struct srl_config {'):txt.index('} srl_config;')]
  fields = re.findall(r'^\s*(?:int32_t|float)\s+([a-z_]+);', body, re.M)
  assert fields == [f[0] for f in CConfig._fields_]
  L = ctypes.CDLL(__import__('stackrl_amd.build', fromlist=['x']).build())
  c = CConfig()
  assert L.srl_config_default(ctypes.byref(c)) == 0
  d = StackConfig(episode_length=30).to_c()
  for name, _ in CConfig._fields_:
    if name == 'max_substeps':     # 0 = "derive it" in the C default; the host mirror hands over int(300 / time_step) (simulator.py:46)
      assert c.max_substeps == 0 and d.max_substeps == 30000
      continue
    assert getattr(c, name) == pytest.approx(getattr(d, name)), name


def test_config_validation_mirrors_reference_errors():
  from stackrl_amd.config import StackConfig
  with pytest.raises(ValueError, match='Invalid value .* for argument dtype'):   # env.py:169-170
    StackConfig(dtype='float33')
  with pytest.raises(ValueError):
    StackConfig(reward_params=-1)                                                # rewarder.py:132-133
  assert StackConfig(rewarder='position').metric_id == 3                          # env.py:148-151
  assert StackConfig(rewarder='occupation').metric_id == 1
  assert StackConfig(rewarder=None).metric_id == 0                                # rewarder.py:113-114
  c = StackConfig.config_gin(episode_length=8).to_c()
  assert c.sim_time_step == pytest.approx(0.0125) and c.metric == 3 and c.reward_scale < 0
  assert StackConfig(resolution_factor=4).n_actions == 2401


def test_product_has_no_oracle_dependency():
  """The product package must not import, include, link or call anything under oracle/."""
  pkg = os.path.join(ROOT, 'stackrl_amd')
  bad = re.compile(r'(import\s+oracle|from\s+oracle|oracle/|oracle\.py|srlo_|libsrl_oracle|srl_oracle)')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.hip', '.h')):
        with open(os.path.join(dirpath, f)) as fh:
          src = fh.read()
        assert not bad.search(src), os.path.join(dirpath, f)


def test_env_requires_gpu_loudly():
  torch = pytest.importorskip('torch')
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from stackrl_amd import env
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    env.VecStackEnv(n_parallel=2)
