"""ctypes loader of libstackrl_hip.so (include/stackrl_hip.h).  There is no CPU fallback: if the HIP
library is missing or cannot be loaded the import of the product path fails loudly."""
import ctypes
import os

from stackrl_amd import build as _build
from stackrl_amd.config import CConfig

_LIB = None

_VP = ctypes.c_void_p
_SIGS = {
  'srl_config_default': (ctypes.c_int, [ctypes.POINTER(CConfig)]),
  'srl_create': (ctypes.c_int, [ctypes.POINTER(CConfig), ctypes.POINTER(_VP)]),
  'srl_destroy': (None, [_VP]),
  'srl_last_error': (ctypes.c_char_p, []),
  'srl_load_meshes': (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, ctypes.c_int32]),
  'srl_seed': (ctypes.c_int, [_VP, ctypes.c_uint32]),
  'srl_set_script': (ctypes.c_int, [_VP, _VP, _VP]),
  'srl_reset': (ctypes.c_int, [_VP, _VP, _VP, _VP]),
  'srl_step': (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
  'srl_sample': (ctypes.c_int, [_VP, _VP, _VP]),
  'srl_sync_status': (ctypes.c_int, [_VP, _VP]),
  'srl_get_state': (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP]),
  'srl_get_maps': (ctypes.c_int, [_VP, _VP, _VP, _VP]),
  'srl_get_velocities': (ctypes.c_int, [_VP, _VP]),
  'srl_get_sweeps': (ctypes.c_int, [_VP, _VP]),
  'srl_set_body_state': (ctypes.c_int, [_VP, _VP, _VP]),
  'srl_step_simulation': (ctypes.c_int, [_VP, ctypes.c_int32, _VP]),
  'srl_get_contacts': (ctypes.c_int, [_VP, _VP, _VP]),
  'srl_render_heightmap': (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
  'srl_get_object_map': (ctypes.c_int, [_VP, ctypes.c_int32, _VP]),
  'srl_set_profiling': (ctypes.c_int, [_VP, ctypes.c_int32]),
  'srl_get_kernel_times': (ctypes.c_int, [_VP, _VP, _VP]),
  'srl_get_order_kernel_times': (ctypes.c_int, [_VP, _VP, _VP]),
  'srl_get_launch_order': (ctypes.c_int, [_VP, _VP, _VP]),
  'srl_set_concurrent_envs': (ctypes.c_int, [_VP, ctypes.c_int32]),
  'srl_set_launch_order': (ctypes.c_int, [_VP, ctypes.c_int32]),
  'srl_build_info': (ctypes.c_char_p, []),
  'srl_get_stage_records': (ctypes.c_int, [_VP, _VP, ctypes.c_int64]),
  'srl_stage_record_stride': (ctypes.c_int32, []),
}
EXPORTS = tuple(sorted(_SIGS))


def path():
  return _build.LIB


def load():
  """Load (building first if the sources are newer) and return the ctypes library."""
  global _LIB
  if _LIB is None:
    if not os.path.isfile(_build.LIB):
      _build.build()
    lib = ctypes.CDLL(_build.LIB)
    for name, (res, args) in _SIGS.items():
      fn = getattr(lib, name)  # AttributeError = symbol missing: fail loudly
      fn.restype = res
      fn.argtypes = args
    _LIB = lib
  return _LIB


def last_error():
  return load().srl_last_error().decode()
