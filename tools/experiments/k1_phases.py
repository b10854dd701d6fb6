"""Where the resident K1 kernel (csrc/nrm_residualize_res.hip) spends an item's time: phase time stamps through nrm_k1_debug_buffer.
Usage: k1_phases.py rows cells dtype(f32|f64) nc"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd.engine import get_engine
from normalisr_amd.association import _prepare_covariates
eng = get_engine()
rows, n, nc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[4])
dt = torch.float64 if sys.argv[3] == 'f64' else torch.float32
g = torch.Generator(device='cuda').manual_seed(5)
x = torch.randn((rows, n), dtype=dt, device='cuda', generator=g)
rp = (rows + 127) // 128 * 128
rng = np.random.default_rng(1)
dc = np.vstack([rng.normal(size=(max(nc - 1, 0), n)), np.ones((1, n))])[:nc] if nc else np.zeros((0, n))
dc64, dci, dcr = _prepare_covariates(dc)
d_c, d_dci = eng.covariates(dc64, dci) if nc else (None, None)
f = lambda: eng.residualize(x, d_c, d_dci, dcr, rows_pad=rp, nslices=6, keep_fp64=False)
f(); f()
seg = 6144 if dt == torch.float32 else 3072
kext = (n + 31) // 32 * 32
groups = (kext + 1023) // 1024
gmax = seg // 1024
nseg = (groups + gmax - 1) // gmax
gseg = (groups + nseg - 1) // nseg
nseg = (groups + gseg - 1) // gseg
items = rp // 4 * nseg
st = torch.zeros((items, 8), dtype=torch.int64, device='cuda')
eng.lib.nrm_k1_debug_buffer(st.data_ptr())
f()
torch.cuda.synchronize()
eng.lib.nrm_k1_debug_buffer(0)
t = st.cpu().numpy().astype(np.float64) / 100.0  # us
names = ['load', 'products', 'wait', 'gather', 'scale', 'digits', 'records']
d = np.diff(t, axis=1)
print('%d items (%d segments per row), kernel span %.1f us' % (items, nseg, t[:, 7].max() - t[:, 0].min()))
print('per item, us: mean / median / p95')
for i, nm in enumerate(names):
	print('  %-9s %7.2f %7.2f %7.2f' % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 95)))
tot = t[:, 7] - t[:, 0]
print('  %-9s %7.2f %7.2f %7.2f' % ('item', tot.mean(), np.median(tot), np.percentile(tot, 95)))
