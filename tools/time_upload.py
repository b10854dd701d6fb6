"""Pageable host -> device upload of a 1 GB fp32 matrix: one copy against chunks of rows (MB per chunk), best of 3."""
import time
import numpy as np, torch
a = np.random.default_rng(0).standard_normal((5000, 50000), dtype=np.float32)
d = torch.empty(a.shape, dtype=torch.float32, device='cuda')
def t(f):
	best = 1e9
	for _ in range(3):
		torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
	return best
x = t(lambda: d.copy_(torch.from_numpy(a)))
print('one copy: %.1f ms (%.1f GB/s)' % (x * 1e3, a.nbytes / x / 1e9))
for mb in (8, 16, 32, 64, 128, 256):
	rows = max(1, mb * (1 << 20) // (a.shape[1] * 4))
	def f():
		for r0 in range(0, a.shape[0], rows):
			d[r0:r0 + rows].copy_(torch.from_numpy(a[r0:r0 + rows]))
	x = t(f)
	print('chunks of %3d MB: %.1f ms (%.1f GB/s)' % (mb, x * 1e3, a.nbytes / x / 1e9))
