import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def ref_pool():
  """69 real reference rocks (64 of the Stack-v0 pool + the five cuboids) as committed data."""
  from stackrl_amd import assets
  return assets.MeshPool.load(os.path.join(GOLDEN, 'ref_rocks.npz'))


@pytest.fixture(scope='session')
def oracle_mod():
  from oracle import oracle
  oracle.build()
  return oracle
