"""Wall-clock of `bin/normalisr coex` on TEXT files (the reference's own file format; BASELINE configs[1] shape by default): the library's
threaded parser / printer against numpy.loadtxt / numpy.savetxt (NRM_TSV=numpy), and the two halves alone.
python tools/time_cli_text.py [genes cells]"""
import os, subprocess, sys, tempfile, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from normalisr_amd import run
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(0)
ng, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 10000)
x = rng.standard_normal((ng, n), dtype=np.float32)
exp, cov, pv, dot = (os.path.join(tmp, f) for f in ('exp.tsv', 'cov.tsv', 'pv.tsv', 'dot.tsv'))
t0 = time.perf_counter()
run.file_write_tsv(exp, x)
print('writing %d x %d as text (%.0f MB): %.2f s' % (ng, n, os.path.getsize(exp) / 1e6, time.perf_counter() - t0), flush=True)
run.file_write_tsv(cov, np.vstack([rng.standard_normal((2, n)), np.ones((1, n))]))
for mode in ('native', 'numpy'):
	os.environ['NRM_TSV'] = mode
	t0 = time.perf_counter()
	y = run.file_read_tsv(exp)
	t1 = time.perf_counter()
	run.file_write_tsv(pv, y[:, :ng] if n >= ng else y)
	t2 = time.perf_counter()
	print('%s: read %.2f s, write of a %d x %d matrix %.2f s' % (mode, t1 - t0, y.shape[0], min(n, ng), t2 - t1), flush=True)
for mode in ('native', 'numpy', 'native'):
	env = dict(os.environ, NRM_TSV=mode)
	t0 = time.perf_counter()
	subprocess.run([os.path.join(root, 'bin', 'normalisr'), 'coex', exp, cov, pv, '--dot_out', dot], check=True, env=env)
	print('normalisr coex on text files, NRM_TSV=%s: %.2f s' % (mode, time.perf_counter() - t0), flush=True)
	if mode == 'native':
		keep = open(pv, 'rb').read()
	elif mode == 'numpy':
		print('P-value files identical:', keep == open(pv, 'rb').read())
