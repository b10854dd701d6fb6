"""cProfile of norm.normvar (numpy in -> numpy out) on a BASELINE configs[1]-sized matrix."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, '.')
import normalisr_amd.normalisr as norm
ng, n, nc = 5000, 10000, 5
rng = np.random.default_rng(1)
dt = rng.standard_normal((ng, n), dtype=np.float32) - 9
dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
w, wt = np.exp(0.25 * rng.normal(size=n)), rng.uniform(0, 1.5, ng)
for _ in range(3):
	norm.normvar(dt, dc, w, wt)
t0 = time.perf_counter(); norm.normvar(dt, dc, w, wt); print('%.1f ms' % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); norm.normvar(dt, dc, w, wt); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
