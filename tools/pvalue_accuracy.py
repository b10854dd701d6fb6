"""Largest relative error of the device's P-value function on the mpmath grid (tests/golden/G12_pvalue_mp.npz), per dof.
Usage: pvalue_accuracy.py   (or through tools/with_lib.py <lib.so> for another build)"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
from normalisr_amd.engine import get_engine
eng = get_engine()
g = np.load('tests/golden/G12_pvalue_mp.npz')
for dof in np.unique(g['dof']):
	m = g['dof'] == dof
	r2, ref, lp = g['r2'][m], g['p'][m], g['log10p'][m]
	d_r2 = torch.from_numpy(r2).cuda()
	d_p = torch.empty_like(d_r2)
	_lib.check(eng.lib.nrm_pvalues_from_r2(d_r2.data_ptr(), r2.size, float(dof), d_p.data_ptr(), 0))
	torch.cuda.synchronize()
	p = d_p.cpu().numpy()
	ok = ref > 1e-300
	err = np.abs(p[ok] / ref[ok] - 1)
	scaled = err / (1e-16 * np.maximum(1.0, np.log(10) * -lp[ok]))
	i = np.argmax(err)
	print('dof %8.0f: max rel err %.2e at R^2 = %.3g (p = %.3g); in units of 1e-16 ln(1/p): %.1f; denormal range abs err %.1e' % (
		dof, err.max(), r2[ok][i], ref[ok][i], scaled.max(), np.abs(p[~ok] - ref[~ok]).max() if (~ok).any() else 0))
