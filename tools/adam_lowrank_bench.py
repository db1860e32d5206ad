#!/usr/bin/env python3
"""Time chaorec_adam_lowrank_f32 (mode 0) and the plain chaorec_adam_step_f32 at a feature-table shape:
python3 tools/adam_lowrank_bench.py [I K R frac]   (default: clothing's image table 11384 x 4096, R 64, 18 % rows hot)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import ops
I, K, R = (int(x) for x in (sys.argv[1:4] or (11384, 4096, 64)))
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.18
dev = torch.device("cuda:0")
torch.manual_seed(0)
p, m, v = torch.randn(I, K, device=dev), torch.zeros(I, K, device=dev), torch.zeros(I, K, device=dev)
W = torch.randn(R, K, device=dev) * 0.1
gy = torch.zeros(I, R, device=dev)
hot = torch.randperm(I, device=dev)[:int(I * frac)]
gy[hot] = torch.randn(hot.numel(), R, device=dev) * 0.05
g = torch.randn(I, K, device=dev) * 0.01
step_dev = torch.ones(1, dtype=torch.int32, device=dev)

def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

t = timed(lambda: ops.adam_lowrank(p, gy, W, m, v, 0, step_dev=step_dev))
gb = 6 * I * K * 4 / 1e9
print(f"adam_lowrank dense  {I}x{K} R={R}: {t:7.1f} us  {gb / t * 1e6 / 1e3:5.2f} TB/s of p,m,v traffic (variant {os.environ.get('CHAOREC_ADAM_VARIANT', '0')})")
t = timed(lambda: ops.adam_step(p, g, m, v, 0, step_dev=step_dev))
print(f"adam_step (dense g) {I}x{K}: {t:7.1f} us  {7 * I * K * 4 / 1e9 / t * 1e6 / 1e3:5.2f} TB/s")
