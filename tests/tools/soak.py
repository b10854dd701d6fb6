#!/usr/bin/env python3
"""Randomised soak of the public API against the CPU oracle (test infrastructure) -- a one-off wider sweep than
tests/test_gpu_parity.py::test_randomised_shapes_vs_oracle: gene counts that exercise every K2 schedule regime and
band cut, odd cell counts, both dtypes, 0-10 covariates.  Usage: soak.py [cases [seed]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import normalisr_amd.normalisr as norm
import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
worst = dict(p=0.0, d=0.0)
from normalisr_amd.engine import get_engine
guard = dict(calls=0, fallbacks=0, worst=0.0)  # the integer engine's accuracy guard over the soak
def note_guard():
	g = get_engine().last_guard
	guard['calls'] += 1
	guard['fallbacks'] += bool(g['fallback'])
	guard['worst'] = max(guard['worst'], g['worst'])
t0 = time.time()
for it in range(cases):
	ng = int(rng.choice([1, 2, 127, 128, 129, 500, 1023, 1025, 1500, 2049, 2600, 3100]))
	n = int(rng.choice([24, 97, 160, 333, 1000, 2001, 2048, 2052, 3001, 4096, 6000]))  # >= 2048 and a multiple of 4: integer Gram engine
	nc = int(rng.integers(0, 11))
	if n <= nc + 3:
		continue
	f32 = bool(rng.integers(2))
	lat = rng.normal(size=(1, n))
	dt = rng.normal(size=(ng, n)) * rng.uniform(0.5, 2, (ng, 1)) + 0.5 * rng.normal(size=(ng, 1)) * lat - 3
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]) if nc else np.zeros((0, n))
	if f32:
		dt = dt.astype(np.float32)
	d64 = dt.astype(np.float64)
	tol = 2e-6 if f32 else 1e-6
	if ng > 1:
		p, d, v = norm.coex(dt, dc)
		note_guard()
		po, do, vo = oracle.coex(d64, dc)
		m = ~np.eye(ng, dtype=bool)
		ep = float(np.max(np.abs(p[m] - po[m]) / np.maximum(po[m], 1e-30)))
		ed = float(np.max(np.abs(d[m] - do[m]) / np.maximum(np.abs(do[m]), 1e-6)))
		assert ep < tol and ed < tol and (p == p.T).all() and (np.diag(p) == 0).all(), ('coex', ng, n, nc, f32, ep, ed)
		worst['p'], worst['d'] = max(worst['p'], ep), max(worst['d'], ed)
	nx = int(rng.choice([1, 3, 33, 140]))
	dg = (rng.random((nx, n)) < 0.3).astype(dt.dtype)
	dg[:, 0], dg[:, 1] = 0, 1  # never constant
	r = norm.de(dg, dt, dc)
	note_guard()
	ro = oracle.de(dg.astype(np.float64), d64, dc)
	ep = float(np.max(np.abs(r[0] - ro[0]) / np.maximum(ro[0], 1e-30)))
	eg = float(np.max(np.abs(r[1] - ro[1]) / np.maximum(np.abs(ro[1]), 1e-6)))
	assert ep < tol and eg < tol, ('de', nx, ng, n, nc, f32, ep, eg)
	worst['p'], worst['d'] = max(worst['p'], ep), max(worst['d'], eg)
	print('case %d ok: genes %d cells %d cov %d %s  (%.0f s)' % (it, ng, n, nc, 'fp32' if f32 else 'fp64', time.time() - t0), flush=True)
print('all ok; worst relative errors', worst, 'guard', guard)
