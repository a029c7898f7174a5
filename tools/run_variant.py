import sys, os, runpy
sys.path.insert(0, os.getcwd())
from stackrl_amd import build as b
if sys.argv[1] != 'product':
  b.LIB = os.path.abspath(sys.argv[1]); b.stale = lambda: False
sys.argv = ['bench_shapes.py'] + sys.argv[2:]
runpy.run_path('tools/bench_shapes.py', run_name='__main__')
