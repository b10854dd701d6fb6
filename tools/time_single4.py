"""norm.de(..., single=4) at BASELINE configs[3] size (1000 gRNAs x 15000 genes x 50000 cells, 5 covariates), numpy -> numpy, with the
engine's per-kernel split.  Usage: time_single4.py [nx ny n] (NRM_GRAM=f64 for the fp64 Gram kernel)."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import normalisr_amd.normalisr as norm
from normalisr_amd.engine import get_engine
nx, ny, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 15000, 50000)
rng = np.random.default_rng(4)
dg = (rng.random((nx, n)) < 0.01).astype(np.float32)
dc = np.vstack([rng.normal(size=(4, n)), np.ones((1, n))])
dt = rng.normal(size=(ny, n)).astype(np.float32)
dt[:16] += 0.2 * dg[0]
eng = get_engine()
norm.de(dg[:64], dt[:256], dc, single=4)
for rep in range(3):
	eng.trace = []
	t0 = time.perf_counter()
	res = norm.de(dg, dt, dc, single=4)
	wall = time.perf_counter() - t0
	split = {}
	for name, e0, e1 in eng.trace:
		split[name] = split.get(name, 0.0) + e0.elapsed_time(e1)
	eng.trace = None
	print('single=4 %d x %d x %d: %.1f ms numpy -> numpy; kernels %s; guard %s' % (nx, ny, n, 1e3 * wall, {k: round(v, 2) for k, v in split.items()}, eng.last_guard))
	del res
t0 = time.perf_counter()
res = norm.de(dg, dt, dc)
print('single=0 for comparison: %.1f ms numpy -> numpy' % (1e3 * (time.perf_counter() - t0)))
