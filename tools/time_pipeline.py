"""Where a pipelined norm.coex call (numpy in -> numpy out) spends its time on the GPU box, plus the raw rates of the copy shapes it
uses (1-D and rectangular device-to-host copies into page-locked memory, pageable host-to-device chunks)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
from normalisr_amd.engine import get_engine
eng = get_engine()
lib = eng.lib
ng, n = 5000, 10000
rng = np.random.default_rng(0)
h = rng.standard_normal((ng, n), dtype=np.float32)
d = torch.empty((ng, ng), dtype=torch.float32, device='cuda')
out = np.empty((ng, ng), dtype=np.float32)
eng.host_pin(out)
st = torch.cuda.current_stream().cuda_stream
def t(f, reps=5):
	f(); torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(reps):
		f()
	torch.cuda.synchronize()
	return (time.perf_counter() - t0) / reps * 1e3
row = ng * 4
print('D2H 1-D 100 MB pinned: %.2f ms' % t(lambda: _lib.check(lib.nrm_copy_to_host(out.ctypes.data, d.data_ptr(), out.nbytes, st))))
print('D2H rect 1024 rows x 20 KB (20 MB): %.2f ms' % t(lambda: _lib.check(lib.nrm_copy_rect_to_host(out.ctypes.data, row, d.data_ptr(), row, 5000 * 4, 1024, st))))
print('D2H rect 4096 rows x 4 KB (16 MB): %.2f ms' % t(lambda: _lib.check(lib.nrm_copy_rect_to_host(out.ctypes.data, row, d.data_ptr(), row, 1024 * 4, 4096, st))))
print('D2H rect 1024 rows x 8 KB (8 MB): %.2f ms' % t(lambda: _lib.check(lib.nrm_copy_rect_to_host(out.ctypes.data, row, d.data_ptr(), row, 2048 * 4, 1024, st))))
eng.host_unpin(out)
print('H2D pageable 40 MB chunk: %.2f ms' % t(lambda: torch.from_numpy(h[:1024]).to('cuda')))
print('H2D pageable 200 MB: %.2f ms' % t(lambda: torch.from_numpy(h).to('cuda')))
import normalisr_amd.normalisr as norm
dc = np.vstack([rng.standard_normal((2, n)), np.ones((1, n))]).astype(np.float32)
norm.coex(h[:256], dc)
for _ in range(3):
	t0 = time.perf_counter(); r = norm.coex(h, dc); print('coex e2e %.2f ms' % ((time.perf_counter() - t0) * 1e3)); r = None
