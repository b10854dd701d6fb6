"""k_s1_stream / k_s1_cells alone at BASELINE configs[3] size (15 000 genes x 50 000 cells fp32, 1000 gRNAs at 0.1 %), with parts of
the work taken away: which part costs what.  python tools/time_single1_stream.py [f32|f64]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
from normalisr_amd import engine as _engine
eng = _engine.get_engine()
lib = eng.lib
f64 = len(sys.argv) > 1 and sys.argv[1] == 'f64'
nx, ny, n = 1000, 15000, 50000
g = torch.Generator(device='cuda').manual_seed(4)
dx = (torch.rand((nx, n), generator=g, device='cuda') < 0.001)
cnt = dx.sum(dim=0)
y = torch.randn((ny, n), generator=g, device='cuda', dtype=torch.float64 if f64 else torch.float32)
ycode = _lib.NRM_F64 if f64 else _lib.NRM_F32
ldye = (ny + 7) // 8 * 8


def run(name, nc, code, reps=5):
	c = torch.randn((max(nc, 1), n), generator=g, device='cuda', dtype=torch.float64)
	n_e = int((code >= 0).sum())
	ye = torch.empty((max(n_e, 1), ldye), dtype=y.dtype, device='cuda')
	common = torch.empty((nc + 1, ny), dtype=torch.float64, device='cuda')
	ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
	for it in range(reps + 2):
		if it == 2:
			ev[0].record()
		_lib.check(lib.nrm_single1_stream(y.data_ptr(), ycode, n, c.data_ptr(), n, nc, code.data_ptr(), n, ny, common.data_ptr(), ye.data_ptr(), ldye, eng._stream()))
	ev[1].record()
	torch.cuda.synchronize()
	ms = ev[0].elapsed_time(ev[1]) / reps
	gb = (ny * n + n_e * ny) * y.element_size() / 1e9
	print('%-46s nc %2d  cells kept %6d  %.3f ms  %.2f TB/s' % (name, nc, n_e, ms, gb / ms))
	return ye, common, n_e


owner = torch.argmax(dx.to(torch.int8), dim=0)
idx_e = torch.nonzero(cnt == 1).flatten()
idx_e = idx_e[torch.argsort(owner[idx_e], stable=True)]
code = torch.where(cnt == 0, -2, -1).to(torch.int32)
code[idx_e] = torch.arange(idx_e.numel(), dtype=torch.int32, device='cuda')  # positions in the order of the groupings, as single1.py
allc = torch.full((n, ), -2, dtype=torch.int32, device='cuda')
run('as a screen runs it', 5, code)
run('no cells kept (sums only)', 5, allc)
run('no covariates', 0, code)
run('no covariates, no cells kept (reads + y^2)', 0, allc)
run('8 covariates (4 rows per workgroup)', 8, code)
run('12 covariates (two passes)', 12, code)
