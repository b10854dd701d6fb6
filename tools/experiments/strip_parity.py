"""The exactness test of tools/experiments/nrm_gram_i8_strip.hip as it ran in tests/test_gpu_round6.py while the strip was part of the library
(round 6; profiles/r06_k2_edge_strip.txt).  Not collected by pytest (no test_ prefix): it needs nrm_gram_i8_strip in the library and gram_i8_impl routing to it."""
import numpy as np
import pytest


@pytest.mark.parametrize('m,n', [(1032, 96), (520, 20000), (1304, 1568)])
def test_integer_gram_ragged_edge_strip_is_exact(m, n, monkeypatch):
	"""Round 6: a symmetric problem whose last 128-row tile holds 1..32 valid rows (5000 genes = 39 x 128 + 8) gets that tile row / column from
	csrc/nrm_gram_i8_strip.hip -- a wave per 32-row block and k-range, operands straight from the digit planes, the pieces added as 64-bit integers and
	rounded once.  Every strip entry is the CORRECTLY ROUNDED exact integer sum of the kept digit products (Python integers, bit for bit), whatever the
	number of cells (1, 8 and 8 pieces here); the rest of the upper triangle is what the one-launch schedule (NRM_DEBUG=k2_strip=0) computes, to the
	last-place differences of tiles the two schedules split differently."""
	import torch
	from normalisr_amd import _lib
	from normalisr_amd.engine import get_engine
	eng = get_engine()
	lib = eng.lib
	ns = 6
	mp, kp = (m + 127) // 128 * 128, (n + 15) // 16 * 16
	assert lib.nrm_gram_i8_strip_bytes(m, kp) > 0 and lib.nrm_gram_i8_strip_bytes(mp, kp) == 0 and lib.nrm_gram_i8_strip_bytes(m + 40, kp) == 0
	rng = np.random.default_rng(700 + m)
	a = np.zeros((mp, kp))
	a[:m, :n] = rng.standard_normal((m, n)) * np.exp(rng.normal(size=(m, 1))) + 0.4 * rng.standard_normal((m, 1)) * rng.standard_normal((1, n))
	st = eng._stream()
	d_a = torch.from_numpy(a).cuda()
	q = torch.empty(int(lib.nrm_quant_bytes(mp, kp, ns)), dtype=torch.uint8, device='cuda')
	ex = torch.empty(mp, dtype=torch.int32, device='cuda')
	_lib.check(lib.nrm_quantize_rows(d_a.data_ptr(), mp, kp, kp, ns, q.data_ptr(), ex.data_ptr(), 0, 0, st))
	work = torch.empty(int(lib.nrm_gram_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
	res = {}
	for switch in ('', 'k2_strip=0'):
		monkeypatch.setenv('NRM_DEBUG', switch)
		dot = torch.full((mp, mp), float('nan'), dtype=torch.float64, device='cuda')
		_lib.check(lib.nrm_gram_i8_band(q.data_ptr(), ex.data_ptr(), 0, q.data_ptr(), ex.data_ptr(), 0, mp, mp, kp, ns, dot.data_ptr(), mp, 1, m, m, 0, mp, work.data_ptr(), st))
		res[switch] = dot.cpu().numpy()
	got, one = res[''], res['k2_strip=0']
	iu = np.triu_indices(m)
	assert np.isfinite(got[iu]).all() and np.isfinite(one[iu]).all()
	scale = np.sqrt(np.outer((a * a).sum(axis=1), (a * a).sum(axis=1)))[:m, :m]
	assert (np.abs(got[:m, :m] - one[:m, :m])[iu] <= 1e-15 * scale[iu]).all()
	e0 = m // 128 * 128
	# the strip against Python integers
	e = ex.cpu().numpy().astype(np.int64)
	qa = np.rint(np.ldexp(a, -e[:, None])).astype(np.int64)
	def digits(v):
		out = []
		for s in range(ns):
			d = v.copy() if s == ns - 1 else ((v & 0xff) ^ 0x80) - 0x80
			v = (v - d) >> 8
			out.append(d)
		return out
	da = digits(qa)
	rows = np.concatenate([rng.integers(0, m, 150), np.arange(e0, m)])
	for i in rows:
		for j in range(e0, m):
			if j < i:
				continue
			exact = 0
			for s in range(ns):
				for u in range(ns):
					if s + u >= ns - 1:
						exact += int(np.dot(da[s][i], da[u][j])) << (8 * (s + u))
			want = float(np.ldexp(float(exact), int(e[i] + e[j])))
			assert got[i, j] == want, (int(i), int(j), float(got[i, j]).hex(), want.hex())
