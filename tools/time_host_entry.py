"""Time nrm_association_tests_host (the pure C entry: host buffers in, host buffers out) on a coex problem.
Usage: time_host_entry.py [genes cells]"""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, '.')
from normalisr_amd import _lib
from normalisr_amd.association import inv_rank

ng, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 10000)
lib = _lib.load()
rng = np.random.default_rng(0)
dt = rng.standard_normal((ng, n), dtype=np.float32)
dc = np.vstack([rng.standard_normal((2, n)), np.ones((1, n))])
dci, rank = inv_rank(dc @ dc.T)
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
for rep in range(4):
	p, d, v = np.empty((ng, ng), np.float32), np.empty((ng, ng), np.float32), np.empty(ng, np.float32)
	t0 = time.perf_counter()
	_lib.check(lib.nrm_association_tests_host(vp(dt), 0, ng, None, 0, 0, vp(dc), 1, 3, n, vp(dci), rank, 0, 1, vp(p), vp(d), None, None, vp(v), None, None, 0))
	print(f'host entry coex {ng}x{n}: {(time.perf_counter() - t0) * 1e3:.1f} ms', flush=True)
