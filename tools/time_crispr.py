#!/usr/bin/env python3
"""Wall-clock of the CRISPR-screen shaped calls (BASELINE configs[3]: 1k gRNAs x genes x 50k cells) through the
public API (numpy in -> numpy out): single=0, single=1 (low MOI), single=4 (high MOI).  Usage: time_crispr.py [genes]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import normalisr_amd.normalisr as norm

ny = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
nx, n = 1000, 50000
rng = np.random.default_rng(4)
lab = rng.integers(0, nx + 200, n)
dg = np.zeros((nx, n), dtype=np.float32)
dg[lab[lab < nx], np.nonzero(lab < nx)[0]] = 1
dc = np.vstack([rng.normal(size=(4, n)), np.ones((1, n))]).astype(np.float32)
dt = rng.standard_normal((ny, n), dtype=np.float32)
dt[:20] += 0.5 * dg[:20]
for single in (0, 1, 4):
	norm.de(dg[:8], dt[:64], dc, single=single)  # warm-up
	times = []
	for rep in range(3):
		t0 = time.perf_counter()
		p = norm.de(dg, dt, dc, single=single)[0]
		times.append(time.perf_counter() - t0)
	dt_s = min(times)
	print('single={}: {} x {} x {} cells: first call {:.3f} s, best of 3 {:.3f} s -> {:.3g} tests/s (min p {:.2g})'.format(
		single, nx, ny, n, times[0], dt_s, nx * ny / dt_s, p.min()), flush=True)
