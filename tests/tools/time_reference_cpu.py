#!/usr/bin/env python3
"""CPU timing of the REFERENCE itself (imported from /root/reference/src, available in the build container only) beside
the oracle port on the same synthetic C2-shaped input (SURVEY 8d 'CPU baseline').  Not used by tests or bench.py.
Usage: OPENBLAS_NUM_THREADS=1 python tools/time_reference_cpu.py [genes cells]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ng, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 10000)
rng = np.random.default_rng(2)
lat = rng.normal(size=(1, n))
dt = rng.normal(size=(ng, n)) + 0.3 * rng.normal(size=(ng, 1)) * lat
dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
pairs = ng * (ng - 1) // 2
cores = os.cpu_count()
import oracle
for nth in (1, cores):
	t0 = time.perf_counter(); po = oracle.coex(dt, dc, nth=nth); t = time.perf_counter() - t0
	print('oracle port     nth=%d: %.1f s  %.3g pairs/s' % (nth, t, pairs / t), flush=True)
sys.path.insert(0, '/root/reference/src')
import warnings
warnings.simplefilter('ignore')
import normalisr.normalisr as ref
for nth in (1, cores):
	t0 = time.perf_counter(); pr = ref.coex(dt, dc, nth=nth); t = time.perf_counter() - t0
	print('reference v1.0.0 nth=%d: %.1f s  %.3g pairs/s' % (nth, t, pairs / t), flush=True)
print('max |p_oracle - p_reference| =', float(np.abs(po[0] - pr[0]).max()))
