"""The parity tests of K1 with its rows resident on chip (tools/experiments/nrm_residualize_res.hip), as they ran in tests/test_gpu_round4.py while
the kernel was part of the shipped library (round 4, opt-in NRM_K1=res).  Parity-green and slower than the two-sweep kernel (4.19 against 3.55 ms
on configs[3] rows: Little's law on the register file, profiles/r04_k1res_phases.txt); it left the library in round 5 with its entry points (the
d_work / work_bytes arguments of nrm_residualize_q*, nrm_residualize_workspace_bytes, nrm_k1_debug_buffer, Engine.k1_work).  """
import numpy as np
import pytest

# archived with its kernel: the entry points it drives are no longer exported by the library (see the header); not collected (the file name
# does not match test_*.py, and the repo-root conftest.py ignores tools/)
pytestmark = pytest.mark.skip(reason="archived experiment: needs the entry points of the kernel restored")

def decode_planes(planes, rows_pad, nks, ns, cks=None):
	"""Fixed-point integers q (rows_pad, 32 nks) from the digit planes of the integer Gram engine: plane s is
	[rows_pad / 32][k-steps][32 rows x 32 bytes], the two 16-byte halves of a row swapped when (row >> 3) & 1 (nrm_gram_i8.hip)."""
	planes = np.asarray(planes).view(np.int8)
	if cks is None:
		cks = nks
	nchunks = nks // cks
	q = np.zeros((rows_pad, nks * 32), dtype=np.int64)
	plane_bytes = (rows_pad // 32) * cks * 1024
	rows = np.arange(rows_pad)
	flip = ((rows & 31) >> 3) & 1
	for c in range(nchunks):
		base = c * ns * plane_bytes
		for s in range(ns):
			img = planes[base + s * plane_bytes: base + (s + 1) * plane_bytes].reshape(rows_pad // 32, cks, 32, 2, 16)
			img = img.transpose(0, 2, 1, 3, 4).reshape(rows_pad, cks, 2, 16)
			img = np.where(flip[:, None, None, None] == 1, img[:, :, ::-1, :], img)
			q[:, c * cks * 32:(c + 1) * cks * 32] += img.reshape(rows_pad, cks * 32).astype(np.int64) << (8 * s)
	return q


def _k1_case(eng, rng, dtype, rows, n, nc, kind):
	import torch
	from normalisr_amd.association import _prepare_covariates
	if kind == 'gauss':
		x = rng.normal(size=(rows, n))
	elif kind == 'mixed':  # rows that are hard for fixed point: large means (loose bound -> the true maximum), sparse, zero
		x = np.vstack([rng.normal(size=(rows - 7, n)), 1e4 + rng.normal(size=(3, n)), (rng.random((3, n)) < 0.01).astype(float), np.zeros((1, n))])
	x = x.astype(dtype)
	if nc:
		dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
		dc64, dci, dcr = _prepare_covariates(dc)
		d_c, d_dci = eng.covariates(dc64, dci)
	else:
		dc64, d_c, d_dci, dcr = np.zeros((0, n)), None, None, 0
	return x, torch.from_numpy(x).cuda(), dc64, d_c, d_dci, dcr


@pytest.mark.parametrize('dtype,rows,n,nc,kind', [
	(np.float32, 200, 2304, 3, 'gauss'),      # one segment per row: no meeting
	(np.float32, 37, 10000, 3, 'mixed'),      # configs[1] rows: 2 segments, padding rows, loose bounds
	(np.float64, 130, 10000, 5, 'mixed'),     # 4 segments, 2.5 covariate passes
	(np.float32, 64, 50000, 5, 'gauss'),      # configs[3] rows: 9 segments
	(np.float64, 9, 500000, 3, 'mixed'),      # configs[4] rows: 163 segments
	(np.float32, 40, 100000, 21, 'gauss'),    # configs[2] covariates
	(np.float64, 12, 30000, 0, 'mixed'),      # no covariates
	(np.float32, 21, 6148, 2, 'gauss'),       # a row one group of 4 cells longer than a segment
])
def test_k1_rows_resident_on_chip(eng, monkeypatch, dtype, rows, n, nc, kind):
	"""K1 keeping its rows on chip between the two phases (one HBM read per row; clusters of workgroups for rows longer than a
	segment) against (a) the fp64 residuals of numpy, (b) the exact-integer host model of the row records, (c) the two-sweep kernel
	k_residualize_v4, (d) itself run again (bitwise), and with its counters left at zero."""
	sys.path.insert(0, TOOLS)
	import i8_error_model as model
	import torch
	rng = np.random.default_rng(401)
	x, d_x, dc64, d_c, d_dci, dcr = _k1_case(eng, rng, dtype, rows, n, nc, kind)
	code = 1 if dtype == np.float64 else 0
	for ns in (6, 5):
		monkeypatch.setenv('NRM_K1', 'res')
		rp = 128 * ((rows + 127) // 128)
		assert eng.k1_work(code, rp, n, nc)[0] != 0, 'the resident kernel does not take this shape'
		r = eng.residualize(d_x, d_c, d_dci, dcr, nslices=ns, keep_fp64=False)
		torch.cuda.synchronize()
		w = list(eng._k1_ws.values())[0][0]
		assert int(w[:16 + 16 * (rp // 4)].view(torch.int32).abs().sum()) == 0, 'K1 left its counters dirty'
		nks = (r.k_pad + 31) // 32
		planes, exps, ss, fix = (t.cpu().numpy() for t in (r._quant[0], r._quant[1], r.ss, r.fix))
		q = decode_planes(planes, r.rows_pad, nks, ns)
		# (d) bitwise reproducible
		r2 = eng.residualize(d_x, d_c, d_dci, dcr, nslices=ns, keep_fp64=False)
		assert torch.equal(r2._quant[0], r._quant[0]) and torch.equal(r2.ss, r.ss) and torch.equal(r2.fix, r.fix) and torch.equal(r2._quant[1], r._quant[1])
		# (a) numpy residuals
		x64 = x.astype(np.float64)
		res = x64 - (x64 @ dc64.T) @ np.linalg.pinv(dc64 @ dc64.T) @ dc64 if nc else x64
		got = np.ldexp(q[:rows, :n].astype(np.float64), exps[:rows, None].astype(np.int64))
		scale = np.ldexp(1.0, exps[:rows].astype(np.int64) + 8 * ns - 2)  # > the row's largest |residual|
		assert (np.abs(res).max(axis=1) <= scale * (1 + 1e-9)).all()
		tol = scale * 2.0**-(8 * ns - 2) * 0.5 + 1e-11 * np.abs(x64).max(axis=1)  # half a unit of the last digit + the rounding of b C
		assert (np.abs(got - res).max(axis=1) <= tol * 1.01).all(), np.abs(got - res).max(axis=1) / tol
		assert (q[rows:] == 0).all() and (q[:, n:] == 0).all(), 'padding rows / cells must carry zero digits'
		assert np.allclose(ss[:rows], (res**2).sum(axis=1), rtol=1e-9, atol=1e-300) and (ss[rows:] == 0).all()
		# (b) the row records against the digits K1 actually wrote
		for i in range(rows):
			d = [t[0] for t in model.digits(q[i:i + 1, :n], ns)]
			st = model.row_stats(d, q[i, :n], exps[i], n, ns)
			assert np.array_equal(fix[i, :ns - 1], np.array(st['u'])), (ns, i)
			assert (fix[i, ns - 1:5] == 0).all()
			if ss[i] > 0:
				assert abs(fix[i, 5] - st['c']) <= 1e-6 * st['c'] + 1e-30 and abs(fix[i, 6] - st['g']) <= 1e-6 * st['g']
				kappa_true = np.abs(res[i]).max() / np.sqrt(ss[i] / n)
				assert fix[i, 7] <= 2.0 * max(12.0, kappa_true) * (1 + 1e-9), (ns, i, fix[i, 7], kappa_true)
			else:
				assert (fix[i] == 0).all()
		# (c) the two-sweep kernel: same exponents, digits equal up to the rounding of b (different summation order)
		monkeypatch.delenv('NRM_K1')
		assert eng.k1_work(code, rp, n, nc) == (0, 0)
		rv = eng.residualize(d_x, d_c, d_dci, dcr, nslices=ns, keep_fp64=False)
		qv = decode_planes(rv._quant[0].cpu().numpy(), rv.rows_pad, nks, ns)
		ev = rv._quant[1].cpu().numpy()
		same = ev[:rows] == exps[:rows]
		assert same.mean() > 0.9, 'fixed-point scales differ between the kernels'
		dq = np.abs(q[:rows][same] - qv[:rows][same]).max(axis=1) if same.any() else np.zeros(1)
		lim = 2 + 1e-11 * np.abs(x64).max(axis=1)[same] / np.ldexp(1.0, exps[:rows][same].astype(np.int64))
		assert (dq <= lim).all(), (dq / lim).max()
		assert np.allclose(rv.ss.cpu().numpy(), ss, rtol=1e-12, atol=0)


def test_k1_resident_chunked_planes_and_coefficients(eng, monkeypatch):
	"""The resident kernel writing cell chunks (what the sharded coex path sends piece by piece: nrm_residualize_q_chunked) and the
	OLS coefficients (alpha, lowmem=False): chunks decode to the same integers as the dense planes; coefficients equal numpy's."""
	import torch
	monkeypatch.setenv('NRM_K1', 'res')
	rng = np.random.default_rng(402)
	rows, n, nc = 70, 20000, 4
	x, d_x, dc64, d_c, d_dci, dcr = _k1_case(eng, rng, np.float32, rows, n, nc, 'gauss')
	r = eng.residualize(d_x, d_c, d_dci, dcr, nslices=6, keep_fp64=False, want_coef=True)
	nks = (r.k_pad + 31) // 32
	q = decode_planes(r._quant[0].cpu().numpy(), r.rows_pad, nks, 6)
	x64 = x.astype(np.float64)
	b = (x64 @ dc64.T) @ np.linalg.pinv(dc64 @ dc64.T)
	assert np.allclose(r.coef.cpu().numpy(), b, rtol=1e-9, atol=1e-12)
	for chunks in (2, 8):
		rc = eng.residualize_chunked(d_x, d_c, d_dci, dcr, 128, 6, chunks)
		cks = rc.cks
		nch = len(rc._quant[0])
		qc = decode_planes(rc._planes.cpu().numpy(), 128, nch * cks, 6, cks=cks)
		assert np.array_equal(qc[:, :nks * 32], q[:128]) and (qc[:, nks * 32:] == 0).all()
		assert torch.equal(rc._quant[1], r._quant[1]) and torch.equal(rc.ss, r.ss) and torch.equal(rc.fix, r.fix)


def test_k1_resident_scratch_reused_across_shapes(eng, monkeypatch):
	"""One scratch serves launches of different shapes on a stream (a de call residualises 1000 design rows, then 15 000 gene rows): the
	partials of a launch on few rows lie where the counters of a launch on more rows must be zero -- the engine zeroes them again.  (Found
	as a hang of `NRM_K1=res bench.py --workload de_c4`.)"""
	import torch
	monkeypatch.setenv('NRM_K1', 'res')
	rng = np.random.default_rng(408)
	n = 20000
	x1, d1, dc64, d_c, d_dci, dcr = _k1_case(eng, rng, np.float32, 100, n, 3, 'gauss')
	x2 = rng.normal(size=(1500, n)).astype(np.float32)
	d2 = torch.from_numpy(x2).cuda()
	want = {}
	for name, d in (('big', d2), ('small', d1)):
		r = eng.residualize(d, d_c, d_dci, dcr, nslices=6, keep_fp64=False)
		want[name] = (r._quant[0].clone(), r.ss.clone(), r.fix.clone())
	for name, d in (('small', d1), ('big', d2), ('small', d1), ('big', d2)):
		r = eng.residualize(d, d_c, d_dci, dcr, nslices=6, keep_fp64=False)
		torch.cuda.synchronize()
		assert torch.equal(r._quant[0], want[name][0]) and torch.equal(r.ss, want[name][1]) and torch.equal(r.fix, want[name][2]), name


