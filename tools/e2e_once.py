"""Three pipelined norm.coex calls (numpy fp32 in -> numpy out, BASELINE configs[1] shape) for a timeline under rocprofv3."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import normalisr_amd.normalisr as norm
rng = np.random.default_rng(0)
ng, n = 5000, 10000
h = rng.standard_normal((ng, n), dtype=np.float32)
dc = np.vstack([rng.standard_normal((2, n)), np.ones((1, n))]).astype(np.float32)
norm.coex(h[:256], dc)
for _ in range(3):
	t0 = time.perf_counter()
	r = norm.coex(h, dc)
	print('coex e2e %.2f ms' % ((time.perf_counter() - t0) * 1e3))
	r = None
