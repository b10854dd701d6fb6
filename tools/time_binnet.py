#!/usr/bin/env python3
"""Device time of binnet (kernel only, p-matrix resident in HBM) on random symmetric p-matrices."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from normalisr_amd import _lib
from normalisr_amd.engine import get_engine
eng = get_engine()
for ng in (5000, 20000):
	g = torch.Generator(device='cuda'); g.manual_seed(1)
	p = torch.rand((ng, ng), generator=g, device='cuda', dtype=torch.float32) ** 3
	p = torch.triu(p, 1); p = p + p.T
	out = torch.empty((ng, ng), dtype=torch.uint8, device='cuda')
	tot = torch.zeros(1, dtype=torch.int64, device='cuda'); fl = torch.zeros(2, dtype=torch.int32, device='cuda')
	for rep in range(3):
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		_lib.check(eng.lib.nrm_binnet(p.data_ptr(), 0, ng, ng, 0.05, out.data_ptr(), ng, tot.data_ptr(), fl.data_ptr(), 0))
		e1.record(); torch.cuda.synchronize()
	ms = e0.elapsed_time(e1)
	print('binnet {0} x {0} fp32: {1:.3f} ms  ({2:.0f} GB/s of p-matrix read + mask written), selected {3}'.format(ng, ms, (ng * ng * 5) / ms / 1e6, int(tot.item())))
