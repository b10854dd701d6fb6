"""cProfile of norm.de (numpy in -> numpy out) at BASELINE configs[3] size: where the wall time of the public call goes."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, '.')
import normalisr_amd.normalisr as norm
ny, nx, n = 15000, 1000, 50000
rng = np.random.default_rng(4)
dg = (rng.random((nx, n)) < 0.01).astype(np.float32)
dc = np.vstack([rng.normal(size=(4, n)), np.ones((1, n))]).astype(np.float32)
dt = rng.standard_normal((ny, n), dtype=np.float32)
norm.de(dg[:8], dt[:64], dc)
norm.de(dg, dt, dc)
t0 = time.perf_counter(); norm.de(dg, dt, dc); print('%.1f ms' % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); norm.de(dg, dt, dc); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
