#!/usr/bin/env python3
"""Randomised soak of single=1 and single=4 against the CPU oracle (test infrastructure): random 0/1 designs with cells carrying
none, one or several groupings, fractional entries, 0-12 covariates, both dtypes.  Usage: soak_single.py [cases [seed]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from normalisr_amd.association import association_tests
import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
worst = {1: 0.0, 4: 0.0}
t0 = time.time()
done = 0
for it in range(cases):
	nx = int(rng.choice([2, 5, 17, 60, 130]))
	ny = int(rng.choice([1, 33, 257, 700]))
	n = int(rng.choice([400, 1501, 4096, 9000]))
	nc = int(rng.integers(0, 13))
	f32 = bool(rng.integers(2))
	none = int(rng.integers(1, nx + 1))  # share of cells without any grouping: none / (nx + none)
	lab = rng.integers(0, nx + none, n)
	dg = np.zeros((nx, n))
	has = lab < nx
	dg[lab[has], np.nonzero(has)[0]] = 1.0
	if has.sum() > 20:
		dbl = rng.choice(np.nonzero(has)[0], min(int(has.sum()) // 10, 50), replace=False)
		dg[rng.integers(0, nx, dbl.size), dbl] = 1.0
	if rng.integers(2):
		dg[0, np.nonzero(lab == 0)[0][::3]] = 0.25
	counts = ((dg != 0) & ((dg != 0).sum(0) == 1)).sum(1)
	if (counts < 3).any() or (~has).sum() < nc + 5:
		continue
	dt = rng.normal(size=(ny, n)) + 0.5 * rng.normal(size=(ny, 1)) * dg[rng.integers(0, nx, ny)]
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]) if nc else np.zeros((0, n))
	x = dt.astype(np.float32) if f32 else dt
	for single in (1, 4):
		if single == 4 and n <= nx + nc + 2:
			continue
		try:
			ref = oracle.association_tests(dg, x.astype(np.float64), dc, single=single, return_dot=False)
		except Exception as e:  # (the reference's own assertions: e.g. a grouping without two distinct values among its cells)
			continue
		out = association_tests(dg, x, dc, single=single, return_dot=False)
		ok = ref[0] > (1e-30 if f32 else 1e-290)
		err = float(np.max(np.abs(out[0][ok] / ref[0][ok] - 1))) if ok.any() else 0.0
		gerr = float(np.max(np.abs(out[1] - ref[1]) / (np.abs(ref[1]) + (1e-5 if f32 else 1e-10))))
		tol = 5e-4 if f32 else 1e-7
		assert err < tol and gerr < (1e-3 if f32 else 1e-6), (it, single, nx, ny, n, nc, f32, err, gerr)
		worst[single] = max(worst[single], err if not f32 else 0.0)
	done += 1
	print('case %d ok: single 1/4, %d groupings x %d genes x %d cells, %d covariates, %s  (%.0f s)' % (it, nx, ny, n, nc, 'fp32' if f32 else 'fp64', time.time() - t0), flush=True)
print('all ok (%d cases run); worst relative P error in fp64:' % done, worst)
