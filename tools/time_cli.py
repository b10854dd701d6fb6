"""Wall-clock of `bin/normalisr coex` on .npy files (BASELINE configs[1] shape), engine path (torch) against NRM_HOST_ENTRY=1 (no torch)."""
import os, subprocess, sys, tempfile, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(0)
ng, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 10000)
np.save(os.path.join(tmp, 'exp.npy'), rng.standard_normal((ng, n), dtype=np.float32))
np.save(os.path.join(tmp, 'cov.npy'), np.vstack([rng.standard_normal((2, n)), np.ones((1, n))]))
for mode in ('0', '1', '0', '1'):
	env = dict(os.environ, NRM_HOST_ENTRY=mode)
	t0 = time.perf_counter()
	subprocess.run([os.path.join(root, 'bin', 'normalisr'), 'coex', os.path.join(tmp, 'exp.npy'), os.path.join(tmp, 'cov.npy'), os.path.join(tmp, 'pv.npy'),
					'--dot_out', os.path.join(tmp, 'dot.npy')], check=True, env=env)
	print('NRM_HOST_ENTRY=%s: %.2f s' % (mode, time.perf_counter() - t0), flush=True)
