"""Worst errors of the integer Gram engine on the shapes of the GPU tests (GPU box).  Prints max |delta r| and the worst relative
error of gamma / covariance against the CPU oracle."""
import os, sys
import numpy as np
sys.path.insert(0, '.')  # run from the repository root
import oracle
from normalisr_amd.association import association_tests
def rel(a, b, floor):
	return float(np.max(np.abs(a - b) / (np.abs(b) + floor)))
for engine in ('i8', 'i8x5', 'f64'):
	os.environ['NRM_GRAM'] = engine
	rng = np.random.default_rng(2020)
	ng, n = 1500, 10000
	dt = rng.normal(size=(ng, n)) * rng.uniform(0.3, 3, (ng, 1)) + 0.3 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n)) + 5
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	po, do, ao, vxo, vo = oracle.association_tests(dt[:300], None, dc)
	res = association_tests(dt, None, dc)
	d = res[1][:300, :300]
	sc = np.sqrt(np.outer(vo, vo))
	print(engine, 'coex 10k cells: max |dr| = %.2e' % float(np.max(np.abs(d - do) / sc)), ' rel(p) = %.2e' % rel(res[0][:300, :300], po, 1e-300))
	os.environ['NRM_DE_PATH'] = 'general'
	rng = np.random.default_rng(303)
	n, nx, ny, nc = 50000, 256, 900, 5
	dc = np.vstack([rng.standard_normal((nc - 1, n)), np.ones((1, n))])
	dg = (rng.random((nx, n)) < 0.01).astype(np.float64)
	dt = rng.standard_normal((ny, n)) + 0.5 * (rng.standard_normal((ny, 6)) @ dg[:6])
	p, gam, a, vg, vt = association_tests(dg, dt, dc, return_dot=False)
	po, go, ao, vgo, vto = oracle.association_tests(dg, dt, dc, return_dot=False)
	r = gam * np.sqrt(vg[:, None] / vt[None, :])
	ro = go * np.sqrt(vgo[:, None] / vto[None, :])
	print(engine, 'de 50k cells:   max |dr| = %.2e' % float(np.max(np.abs(r - ro))), ' worst rel(gamma, floor 1e-12) = %.2e' % rel(gam, go, 1e-12),
		  ' rel(p) = %.2e' % rel(p, po, 1e-300), ' min |gamma| = %.1e' % float(np.abs(go).min()))
	del os.environ['NRM_DE_PATH']
