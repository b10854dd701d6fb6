"""Where a resident single=1 call (BASELINE configs[3] shape, gRNA incidence 0.1 %) spends its wall time: cProfile of three steps."""
import cProfile
import pstats
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd.single1 import association_tests_single1 as fn
nx, ny, n = 1000, 15000, 50000
g = torch.Generator(device='cuda').manual_seed(4)
dc = torch.cat([torch.randn((4, n), generator=g, device='cuda'), torch.ones((1, n), device='cuda')]).cpu().numpy().astype(np.float64)
dx = (torch.rand((nx, n), generator=g, device='cuda') < 0.001).float()
dy = torch.randn((ny, n), generator=g, device='cuda')
fn(dx, dy, dc, return_dot=False, device_out=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
	out = fn(dx, dy, dc, return_dot=False, device_out=True)
torch.cuda.synchronize()
print('%.1f ms per step' % (1e3 * (time.perf_counter() - t0) / 3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
	out = fn(dx, dy, dc, return_dot=False, device_out=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
