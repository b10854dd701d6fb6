#!/usr/bin/env python3
"""Device time of binnet (kernel only, p-matrix resident in HBM) on random symmetric p-matrices.
Usage: time_binnet.py [ng dtype(f32|f64) power] ...   (p = rand ** power: power 1 = null P-values, 3 = a dense network)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from normalisr_amd import _lib
from normalisr_amd.engine import get_engine
eng = get_engine()
cases = [(5000, 'f32', 3), (20000, 'f32', 3), (20000, 'f32', 1), (30000, 'f64', 1), (30000, 'f64', 3)]
if len(sys.argv) > 3:
	cases = [(int(sys.argv[1]), sys.argv[2], float(sys.argv[3]))]
for ng, dt, power in cases:
	g = torch.Generator(device='cuda'); g.manual_seed(1)
	tdt = torch.float32 if dt == 'f32' else torch.float64
	p = torch.rand((ng, ng), generator=g, device='cuda', dtype=tdt) ** power
	p[:, :64] *= 1e-6
	p = torch.triu(p, 1); p = p + p.T
	out = torch.empty((ng, ng), dtype=torch.uint8, device='cuda')
	tot = torch.zeros(1, dtype=torch.int64, device='cuda'); fl = torch.zeros(2, dtype=torch.int32, device='cuda')
	for rep in range(3):
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		_lib.check(eng.lib.nrm_binnet(p.data_ptr(), 0 if dt == 'f32' else 1, ng, ng, 0.05, out.data_ptr(), ng, tot.data_ptr(), fl.data_ptr(), 0))
		e1.record(); torch.cuda.synchronize()
	ms = e0.elapsed_time(e1)
	st = torch.zeros((ng, 6), dtype=torch.int64, device='cuda')
	eng.lib.nrm_binnet_debug_buffer(st.data_ptr())
	_lib.check(eng.lib.nrm_binnet(p.data_ptr(), 0 if dt == 'f32' else 1, ng, ng, 0.05, out.data_ptr(), ng, tot.data_ptr(), fl.data_ptr(), 0))
	torch.cuda.synchronize()
	eng.lib.nrm_binnet_debug_buffer(0)
	passes = st[:, 4].float().mean().item()
	t = st[:, :4].cpu().numpy().astype(float) / 100.0
	d = t[:, 1:] - t[:, :-1]
	print('   per row, us (mean): load+check %.2f, threshold passes %.2f, mask %.2f; %.1f counting passes; row %.2f; kernel span %.1f us' % (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), passes, (t[:, 3] - t[:, 0]).mean(), t[:, 3].max() - t[:, 0].min()))
	esz = 4 if dt == 'f32' else 8
	print('binnet {0} x {0} {4} (rand^{5:g}): {1:.3f} ms  ({2:.0f} GB/s of p-matrix read + mask written), selected {3}'.format(ng, ms, (ng * ng * (esz + 1)) / ms / 1e6, int(tot.item()), dt, power))
	del p, out
