import sys, os, subprocess
ROOT = '/root/repo'
sys.path.insert(0, ROOT)
from stackrl_amd import build as B
flag = sys.argv[1]
so = os.path.join(ROOT, 'gpurun_out', 'librx.so')
subprocess.check_call(['/opt/rocm/bin/hipcc'] + B.FLAGS + ([flag] if flag != 'none' else []) + [os.path.join(B.CSRC, 'stackrl_hip.hip'), '-o', so], stderr=subprocess.DEVNULL)
B.LIB = so
sys.argv = ['x']
exec(open(os.path.join(ROOT, 'tools/bench_render.py')).read())
