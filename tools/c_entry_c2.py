"""BASELINE configs[1] (coex 5000 genes x 10 000 cells fp32, 3 covariates) numpy in -> numpy out through the package's torch engine (uploads, kernels and
copy-out pipelined chunk by chunk) and through the library's whole-problem entry (NRM_HOST_ENTRY=1): wall time per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import normalisr_amd.normalisr as norm
rng = np.random.default_rng(2)
ng, n = 5000, 10000
dt = rng.standard_normal((ng, n), dtype=np.float32)
dc = np.vstack([rng.standard_normal((2, n)), np.ones((1, n))])
res = {}
for route in ('package', 'c_entry'):
	os.environ['NRM_HOST_ENTRY'] = '1' if route == 'c_entry' else '0'
	norm.coex(dt[:256], dc)
	ts = []
	for _ in range(6):
		t0 = time.perf_counter()
		out = norm.coex(dt, dc)
		ts.append(time.perf_counter() - t0)
	print(route, ' '.join('%.1f' % (t * 1e3) for t in ts))
	res[route] = (min(ts), out)
	out = None
a, b = res['package'][1][0].astype(np.float64), res['c_entry'][1][0].astype(np.float64)
ok = a > 1e-30
print('coex %d x %d: package %.1f ms   C entry %.1f ms   largest relative difference of P: %.2e' % (ng, n, res['package'][0] * 1e3, res['c_entry'][0] * 1e3, float(np.max(np.abs(b[ok] / a[ok] - 1)))))
