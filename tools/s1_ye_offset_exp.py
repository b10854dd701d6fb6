"""Round 6: the two step times of the resident single=1 step (1.61 / 1.75 ms from run to run of one binary) -- does the placement of the stream kernel's transposed
output relative to the expression matrix decide?  One process, one plan; the output buffer moved through a larger allocation in steps; k_s1_stream timed at each offset."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd.single1 import Single1Plan
from normalisr_amd import _lib

nx, ny, n, nc = 1000, 15000, 50000, 5
g = torch.Generator(device='cuda').manual_seed(4)
dc = torch.cat([torch.randn((nc - 1, n), generator=g, device='cuda'), torch.ones((1, n), device='cuda')])
dx = (torch.rand((nx, n), generator=g, device='cuda') < 0.001).to(torch.float32)
dy = torch.randn((ny, n), generator=g, device='cuda')
plan = Single1Plan(dx, dy, dc.cpu().numpy().astype(np.float64), return_dot=False)
plan.step()
eng, lib = plan.eng, plan.eng.lib
rows = plan.ye.shape[0]
big = torch.empty((rows * plan.ldye + (64 << 20) // 4, ), dtype=torch.float32, device='cuda')
print('Y at %#x, YE (plan) at %#x, big at %#x; %d kept cells' % (dy.data_ptr(), plan.ye.data_ptr(), big.data_ptr(), rows))


def time_stream(ye):
	st = eng._stream()
	def run():
		_lib.check(lib.nrm_single1_stream(dy.data_ptr(), _lib.NRM_F32, dy.stride(0), plan.d_c.data_ptr(), n, nc, plan.code.data_ptr(), n, ny, plan.common.data_ptr(), ye.data_ptr(), plan.ldye, st))
	for _ in range(3):
		run()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(10):
		run()
	e1.record()
	torch.cuda.synchronize()
	return e0.elapsed_time(e1) / 10


print('plan buffer: %.3f ms' % time_stream(plan.ye))
for off in (0, 64, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 1 << 22, 1 << 24, 3 << 23):
	ye = big[off // 4:off // 4 + rows * plan.ldye].view(rows, plan.ldye)
	print('offset %9d B (address %#x): %.3f ms' % (off, ye.data_ptr(), time_stream(ye)))
