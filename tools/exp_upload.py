"""Experiment: host -> device of a 3 GB numpy array (the expression matrix of BASELINE configs[3]): torch's pageable copy against page-locking the
user's array for the duration of the copy (nrm_host_pin / nrm_copy... the way results are downloaded)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import engine as _engine, _lib
eng = _engine.get_engine()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
a = np.random.default_rng(0).standard_normal((rows, 50000), dtype=np.float32)
d = torch.empty(a.shape, dtype=torch.float32, device='cuda')
for name in ('pageable', 'pinned for the copy', 'staged (nrm_upload)', 'pageable', 'pinned for the copy', 'staged (nrm_upload)', 'staged (nrm_upload)'):
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	if name.startswith('staged'):
		_lib.check(eng.lib.nrm_upload(a.ctypes.data, d.data_ptr(), a.nbytes, 0, eng._stream()))
		t1 = time.perf_counter()
		torch.cuda.synchronize()
		t2 = time.perf_counter()
		ok = bool((d[::977, ::31].cpu() == torch.from_numpy(a[::977, ::31])).all()) and bool((d[-1].cpu() == torch.from_numpy(a[-1])).all())
		assert ok
	elif name == 'pageable':
		d.copy_(torch.from_numpy(a))
		torch.cuda.synchronize()
		t1 = t2 = time.perf_counter()
	else:
		eng.host_pin(a)
		t1 = time.perf_counter()
		d.copy_(torch.from_numpy(a), non_blocking=True)
		torch.cuda.synchronize()
		t2 = time.perf_counter()
		eng.host_unpin(a)
	t3 = time.perf_counter()
	print('%-22s %.1f ms (pin %.1f, copy %.1f, unpin %.1f) = %.1f GB/s' % (name, 1e3 * (t3 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), a.nbytes / (t3 - t0) / 1e9), flush=True)
