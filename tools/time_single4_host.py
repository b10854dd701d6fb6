"""Where a resident single=4 call (BASELINE configs[3] size) spends its wall time: every engine upload / download timed with its
size, plus cProfile of three steps."""
import cProfile
import pstats
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd.single4 import association_tests_single4 as fn
from normalisr_amd.engine import get_engine, Engine
nx, ny, n = 1000, 15000, 50000
g = torch.Generator(device='cuda').manual_seed(4)
dc = torch.cat([torch.randn((4, n), generator=g, device='cuda'), torch.ones((1, n), device='cuda')]).cpu().numpy().astype(np.float64)
dx = (torch.rand((nx, n), generator=g, device='cuda') < 0.01).float()
dy = torch.randn((ny, n), generator=g, device='cuda')
fn(dx, dy, dc, return_dot=False, device_out=True)
eng = get_engine()
log = []
orig = Engine.upload.__wrapped__ if hasattr(Engine.upload, '__wrapped__') else Engine.upload


def timed_upload(self, a, dtype=None):
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	r = orig(self, a, dtype)
	torch.cuda.synchronize()
	log.append(('upload', np.asarray(a).nbytes, 1e3 * (time.perf_counter() - t0)))
	return r


Engine.upload = timed_upload
t0 = time.perf_counter()
fn(dx, dy, dc, return_dot=False, device_out=True)
torch.cuda.synchronize()
print('one step with synchronised uploads: %.1f ms' % (1e3 * (time.perf_counter() - t0)))
for kind, nb, ms in log:
	print('  %s %10d bytes %8.3f ms' % (kind, nb, ms))
Engine.upload = orig if not hasattr(Engine.upload, '__wrapped__') else Engine.upload
import os
os.environ.pop('NRM_S4_TRACE', None)
for rep in range(3):
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(5):
		out = fn(dx, dy, dc, return_dot=False, device_out=True)
	torch.cuda.synchronize()
	print('5 steps back to back: %.1f ms per step' % (1e3 * (time.perf_counter() - t0) / 5))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
	out = fn(dx, dy, dc, return_dot=False, device_out=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
