#!/usr/bin/env python3
"""Per-kernel breakdown of the steady-state Q-net rollout forward.

Run under `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/profile_qnet.py run [B] [dtype]`,
then `python3 tools/profile_qnet.py parse <dir>`: kernels between the two marker launches (arange of 12,345 / 23,456
elements) are the timed iterations; warm-up and MIOpen's solver search are excluded."""
import sys, os, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ITERS = 3


def run(B, dtype):
  import torch
  from stackrl_amd import nets, qops
  net = nets.DeepQSiamFCN(seed=1).cuda().eval()
  g = torch.Generator(device='cuda').manual_seed(0)
  xm = torch.randint(0, 256, (B, 128, 128, 2), generator=g, device='cuda', dtype=torch.uint8)
  xo = torch.randint(0, 256, (B, 32, 32, 1), generator=g, device='cuda', dtype=torch.uint8)
  ac = {'fp32': None, 'fp32-stock': None, 'bf16': torch.bfloat16, 'fp16': torch.float16}[dtype]
  if ac is not None:
    net = net.to(memory_format=torch.channels_last)
  pol = qops.FusedPolicy(chunk=min(B, 512), autocast=ac, fast=(dtype != 'fp32-stock'))
  with torch.no_grad():
    for _ in range(3): pol(net, (xm, xo), 0.1, g)
    torch.cuda.synchronize()
    torch.arange(12345, device='cuda'); torch.cuda.synchronize()
    for _ in range(ITERS): pol(net, (xm, xo), 0.1, g)
    torch.cuda.synchronize()
    torch.arange(23456, device='cuda'); torch.cuda.synchronize()


def parse(d):
  f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
  rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
  marks = [i for i, r in enumerate(rows) if 'arange' in r['Kernel_Name']]
  a, b = marks[-2], marks[-1]
  agg = {}
  for r in rows[a + 1:b]:
    n = r['Kernel_Name']; dt = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    c = agg.setdefault(n, [0, 0]); c[0] += dt; c[1] += 1
  tot = sum(v[0] for v in agg.values())
  print('steady state: %.2f ms per forward in kernels, %d launches per forward' % (tot / ITERS / 1e6, (b - a - 1) / ITERS))
  for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):          # every kernel of the window, not a top list
    print('%6.2f%%  %9.1f us avg  %6.1f calls/fwd  %s' % (100 * t / tot, t / c / 1e3, c / ITERS, n[:140]))
  mine = sum(v[0] for n, v in agg.items() if 'k_' in n or 'srl_' in n)
  print('kernels of this repo (k_* / srl_*): %.2f %% of the kernel time, %d of %d launches per forward' % (
    100.0 * mine / tot, sum(v[1] for n, v in agg.items() if 'k_' in n or 'srl_' in n) / ITERS, (b - a - 1) / ITERS))


if __name__ == '__main__':
  if sys.argv[1] == 'run':
    run(int(sys.argv[2]) if len(sys.argv) > 2 else 512, sys.argv[3] if len(sys.argv) > 3 else 'bf16')
  else:
    parse(sys.argv[2])
