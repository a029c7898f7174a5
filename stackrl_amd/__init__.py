"""stackrl_amd — MI355X-native batched Stack-v0 env step.

`from stackrl_amd import envs; envs.make('Stack-v0', n_parallel=1024, seed=11)` mirrors
`stackrl.envs.make` (stackrl/envs/utils.py:44).  Importing the package does not touch the GPU; creating an
env loads libstackrl_hip.so and fails loudly if it is missing.
"""
from stackrl_amd import config  # noqa: F401
from stackrl_amd.config import StackConfig  # noqa: F401

__all__ = ['config', 'StackConfig', 'assets', 'envs']


def __getattr__(name):
  if name == 'envs':
    import stackrl_amd.env as envs
    return envs
  if name == 'assets':
    import stackrl_amd.assets as assets
    return assets
  raise AttributeError(name)
