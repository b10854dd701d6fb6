"""BASELINE configs[2] at full size (1 grouping x 20 000 genes x 100 000 cells fp32, 20 covariates) numpy in -> numpy out through the package (torch engine: the
streaming de kernel) and through the library's whole-problem entry (NRM_HOST_ENTRY=1): wall time per call, the 8 GB upload included."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import normalisr_amd.normalisr as norm
ny, n, nc = 20000, 100000, 20
rng = np.random.default_rng(5)
dt = rng.standard_normal((ny, n), dtype=np.float32)
dt += 9
dc = np.vstack([rng.standard_normal((nc - 1, n)), np.ones((1, n))])
dg = (rng.random((1, n)) < 0.5).astype(np.float32)
dt[:50] += 0.05 * dg[0]
res = {}
for route in ('package', 'c_entry'):
	os.environ['NRM_HOST_ENTRY'] = '1' if route == 'c_entry' else '0'
	norm.de(dg, dt[:64], dc)
	best = 1e9
	for _ in range(3):
		t0 = time.perf_counter()
		out = norm.de(dg, dt, dc)
		best = min(best, time.perf_counter() - t0)
	res[route] = (best, out)
p0, p1 = res['package'][1][0].astype(np.float64), res['c_entry'][1][0].astype(np.float64)
ok = p0 > 1e-30
print('de 1 x %d x %d, %d covariates: package %.1f ms   C entry %.1f ms   largest relative difference of P: %.2e' % (
	ny, n, nc, res['package'][0] * 1e3, res['c_entry'][0] * 1e3, float(np.max(np.abs(p1[ok] / p0[ok] - 1)))))
