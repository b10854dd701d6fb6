"""Do host-to-device and device-to-host copies overlap on this box?  Times 200 MB each way alone and together (two streams),
for the copy flavours the pipelined coex uses: pageable / page-locked uploads, 1-D and rectangular downloads."""
import sys, time, threading
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
from normalisr_amd.engine import get_engine
eng = get_engine()
lib = eng.lib
ng, n = 5000, 10000
h = np.random.default_rng(0).standard_normal((ng, n), dtype=np.float32)
hp = h.copy()
eng.host_pin(hp)
d_in = torch.empty((ng, n), dtype=torch.float32, device='cuda')
d_out = torch.randn((2 * ng, ng), dtype=torch.float32, device='cuda')
out = np.empty((2 * ng, ng), dtype=np.float32)
eng.host_pin(out)
s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()


def up(src):
	with torch.cuda.stream(s_up):
		d_in.copy_(torch.from_numpy(src), non_blocking=True)


def down_1d():
	_lib.check(lib.nrm_copy_to_host(out.ctypes.data, d_out.data_ptr(), out.nbytes, s_dn.cuda_stream))


def down_rect():
	row = ng * 4
	for i in range(10):  # ten 1000-row rectangles of half the row width each, twice
		for half in (0, 1):
			_lib.check(lib.nrm_copy_rect_to_host(out.ctypes.data + i * 1000 * row + half * row // 2, row, d_out.data_ptr() + i * 1000 * row + half * row // 2, row,
												 row // 2, 1000, s_dn.cuda_stream))


def t(fs, reps=5):
	def once():
		th = [threading.Thread(target=f) for f in fs[1:]]
		for x in th:
			x.start()
		fs[0]()
		for x in th:
			x.join()
		torch.cuda.synchronize()
	once()
	t0 = time.perf_counter()
	for _ in range(reps):
		once()
	return (time.perf_counter() - t0) / reps * 1e3


print('H2D 200 MB pageable %.2f ms, page-locked %.2f ms' % (t([lambda: up(h)]), t([lambda: up(hp)])))
print('D2H 200 MB 1-D %.2f ms, 100 MB as 20 rectangles %.2f ms' % (t([down_1d]), t([down_rect])))
print('together: pageable up + 1-D down %.2f ms; page-locked up + 1-D down %.2f ms' % (t([lambda: up(h), down_1d]), t([lambda: up(hp), down_1d])))
print('together: pageable up + rect down %.2f ms; page-locked up + rect down %.2f ms' % (t([lambda: up(h), down_rect]), t([lambda: up(hp), down_rect])))
