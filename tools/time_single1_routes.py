"""Round 6, verdict item 1(d): where a resident single=1 step spent its time when the groupings' statistics were finished on the host (rounds 3-5:
NRM_DEBUG=single1_stats=host keeps that route), under whatever BLAS thread setting the environment gives -- beside the device route (Single1Plan).
configs[3] size, inputs resident in HBM.  Usage: time_single1_routes.py [calls]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from normalisr_amd.single1 import Single1Plan, association_tests_single1

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nx, ny, n, nc = 1000, 15000, 50000, 5
g = torch.Generator(device='cuda').manual_seed(4)
dc = torch.cat([torch.randn((nc - 1, n), generator=g, device='cuda'), torch.ones((1, n), device='cuda')])
dx = (torch.rand((nx, n), generator=g, device='cuda') < 0.001).to(torch.float32)
dy = torch.randn((ny, n), generator=g, device='cuda')
dc_h = dc.cpu().numpy().astype(np.float64)


def timed(fn):
	for _ in range(3):
		fn()
	torch.cuda.synchronize()
	ts = []
	for _ in range(calls):
		t0 = time.perf_counter()
		fn()
		torch.cuda.synchronize()
		ts.append(1e3 * (time.perf_counter() - t0))
	return min(ts), float(np.median(ts)), max(ts)


os.environ['NRM_DEBUG'] = 'single1_stats=host'
host = timed(lambda: association_tests_single1(dx, dy, dc_h, return_dot=False, device_out=True))
os.environ['NRM_DEBUG'] = 'single1_stats=host,s1_trace=1'
import logging
logging.basicConfig(level=logging.WARNING)
association_tests_single1(dx, dy, dc_h, return_dot=False, device_out=True)  # one call with the phases printed
del os.environ['NRM_DEBUG']
plan = Single1Plan(dx, dy, dc_h, return_dot=False)
dev = timed(plan.step)
plan.check()
print('threads OPENBLAS=%s OMP=%s, %d host cores: statistics on the host (rounds 3-5) min / median / max %.2f / %.2f / %.2f ms per call; on the device (Single1Plan, HIP graph) %.2f / %.2f / %.2f' % (
	os.environ.get('OPENBLAS_NUM_THREADS', 'unset'), os.environ.get('OMP_NUM_THREADS', 'unset'), os.cpu_count(), *host, *dev))
