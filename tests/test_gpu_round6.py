"""GPU tests added in round 6: the resident single=1 and single=4 steps with nothing on the host (Single1Plan, Single4Plan: the groupings' pseudo-inverses,
integer ranks and P-value plans on the device, the Newton-Schulz inverse with the step count of the first call, one HIP graph per step), the torch-free
single=1 entry on the same kernels, device selection on the library's whole-problem entries.  Same parity bar as test_gpu_parity.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from conftest import ROOT
from test_gpu_parity import close, p_close

pytestmark = pytest.mark.gpu


def _screen(rng, nx, ny, n, nc, moi=1.0, valued=False, dup_cov=False):
	dx = (rng.random((nx, n)) < moi / nx).astype(np.float64)
	if valued:
		dx[1] *= rng.uniform(0.5, 1.0, n)
		dx[1, np.nonzero(dx[1])[0][0]] = 1.0
	dy = rng.normal(size=(ny, n))
	dy[:5] += 0.5 * dx[0]
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]) if nc else np.zeros((0, n))
	if dup_cov and nc >= 3:
		dc[1] = dc[0]  # a covariate twice: every grouping's C_S C_S^T is rank deficient (the integer rank decides dof)
	return dx, dy, dc


@pytest.mark.parametrize('nc,valued,dup,dtype', [(0, False, False, np.float64), (3, True, False, np.float64), (5, False, True, np.float32), (8, True, False, np.float64)])
def test_single1_plan_keeps_the_host_out_of_a_step(nc, valued, dup, dtype):
	"""Single1Plan against the oracle's per-grouping loop (association.py:263-390,911-925): pseudo-inverse, INTEGER rank (a covariate given twice: rank nc - 1,
	dof one more), ccx, vx, dof and the P-value plan of every grouping come from k_s1_group_info.  Steps 3.. are replays of one HIP graph -- a step
	that touched the host could not have been captured -- and give the first step's bits."""
	import torch
	from normalisr_amd.single1 import Single1Plan, association_tests_single1
	rng = np.random.default_rng(600 + nc)
	nx, ny, n = 40, 70, 6000
	dx, dy, dc = _screen(rng, nx, ny, n, nc, valued=valued, dup_cov=dup)
	dy = dy.astype(dtype)
	want = oracle.association_tests(dx, dy.astype(np.float64), dc, single=1, return_dot=False, lowmem=False)
	plan = Single1Plan(torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), dc, return_dot=False, lowmem=False)
	plan.step()
	first = plan.results()
	tol = 1e-6 if dtype == np.float32 else 1e-9
	assert first[0].dtype == dtype and p_close(first[0], want[0], 1e-6) and close(first[1], want[1], tol, 1e-12) and close(first[3], want[3], tol) and close(first[4], want[4], tol)
	if nc:
		assert close(first[2], want[2], max(tol, 1e-8), 1e-9)
	for _ in range(4):
		plan.step()
	assert plan._graph.graph is not None  # captured: no synchronisation, read-back or upload inside a step
	again = plan.results()
	for a, b in zip(first, again):
		assert (a is None and b is None) or np.array_equal(a, b)
	# the public call takes the same route (numpy in) and the round-5 route with the statistics on the host agrees with it
	got = association_tests_single1(dx, dy, dc, return_dot=False, lowmem=False)
	assert all((a is None and b is None) or np.array_equal(a, b) for a, b in zip(first, got))
	os.environ['NRM_DEBUG'] = 'single1_stats=host'
	try:
		host = association_tests_single1(dx, dy, dc, return_dot=False, lowmem=False)
	finally:
		del os.environ['NRM_DEBUG']
	assert p_close(host[0], first[0], 1e-9) and close(host[1], first[1], 1e-9, 1e-12) and close(host[3], first[3], 1e-9)


def test_single1_plan_raises_what_the_reference_raises():
	"""The checks the host made between the kernels are counters now, read once in results(): a grouping with a single value on its selected cells
	(association.py:917-918: AssertionError), too few cells for the degrees of freedom removed (RuntimeError), non-finite covariates (the SVD's ValueError)."""
	import torch
	from normalisr_amd.single1 import Single1Plan, association_tests_single1
	rng = np.random.default_rng(611)
	nx, ny, n, nc = 12, 30, 4100, 3
	dx, dy, dc = _screen(rng, nx, ny, n, nc)
	one = dx.copy()
	one[:] = 0
	for i in range(nx):
		one[i, i::nx] = 1  # every cell carries exactly one grouping: no shared cells, grouping i is constant 1 on its own cells
	with pytest.raises(AssertionError):
		association_tests_single1(one, dy, dc)
	with pytest.raises(AssertionError):
		oracle.association_tests(one, dy, dc, single=1)
	with pytest.raises(RuntimeError):
		association_tests_single1(dx, dy, dc, dimreduce=n)
	bad = dc.copy()
	bad[0, 7] = np.nan
	with pytest.raises(ValueError):
		association_tests_single1(dx, dy, bad)
	# and a plan stays usable after a failed look: the counters are cleared by it
	plan = Single1Plan(torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), dc, dimreduce=n)
	plan.step()
	with pytest.raises(RuntimeError):
		plan.results()
	plan.dimreduce = 0
	plan._graph.graph, plan._graph.calls = None, 0
	plan.step()
	want = oracle.association_tests(dx, dy, dc, single=1)
	assert p_close(plan.results()[0], want[0])
	# the design written to IN PLACE between steps (a captured graph and entry lists of the old values): the plan notices (._version) and builds itself anew
	for _ in range(3):
		plan.step()
	assert plan._graph.graph is not None
	d_x = plan.d_dx
	free = torch.nonzero(d_x.sum(dim=0) == 0).flatten()[:40]
	d_x[3, free] = 1.0  # forty more cells for grouping 3, taken from the shared ones
	plan.step()
	dx2 = d_x.cpu().numpy()
	want2 = oracle.association_tests(dx2, dy, dc, single=1)
	got2 = plan.results()
	assert plan.n_kept == int(((dx2 != 0).sum(axis=0) == 1).sum()) and p_close(got2[0], want2[0]) and not np.array_equal(got2[0][3], want[0][3])


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_single4_plan_replays_the_public_call(dtype, monkeypatch):
	"""Single4Plan: the first step IS association_tests_single4 (rank certificate, entry lists, the Newton-Schulz start and step count); later steps are
	that call's device work as one HIP graph -- the same bits -- and the oracle's per-grouping SVD loop (association.py:421-576) agrees.  Inputs written to
	between steps, or a step whose counters do not stand, go back through the public call."""
	import torch
	from normalisr_amd.single4 import Single4Plan, association_tests_single4
	monkeypatch.setenv('NRM_DE_SPARSE', 'force')  # (the size rule would leave so small a screen to K1 + K2)
	rng = np.random.default_rng(640)
	nx, ny, n, nc = 48, 90, 6000, 4
	dx = (rng.random((nx, n)) < 0.02).astype(dtype)
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
	dy = (rng.normal(size=(ny, n)) + 0.4 * dx[3] + 0.3 * dc[0]).astype(dtype)
	want = oracle.association_tests(dx.astype(np.float64), dy.astype(np.float64), dc, single=4, return_dot=False)
	d_x, d_y = torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda()
	ref = association_tests_single4(d_x, d_y, dc, return_dot=False)
	tol = 1e-6 if dtype == np.float32 else 1e-9
	assert p_close(ref[0], want[0], 1e-6) and close(ref[1], want[1], tol, 1e-12) and close(ref[3], want[3], tol) and close(ref[4], want[4], tol)
	plan = Single4Plan(d_x, d_y, dc, return_dot=False)
	plan.step()
	assert plan.lean is True
	first = plan.results()
	for _ in range(4):
		plan.step()
	assert plan._graph.graph is not None and plan.check() and plan.fallbacks == 0
	lean = plan.results()
	for a, b, c in zip(ref, first, lean):
		assert (a is None and b is None and c is None) or (np.array_equal(a, b) and np.array_equal(a, c))
	# the expression matrix written to in place: the next step is the public call again (and decides anew), results follow the new values
	d_y[7] += 2.0 * d_x[5]
	plan.step()
	changed = plan.results()
	dy2 = d_y.cpu().numpy()
	want2 = oracle.association_tests(dx.astype(np.float64), dy2.astype(np.float64), dc, single=4, return_dot=False)
	ok = want2[0] > (1e-30 if dtype == np.float32 else 1e-290)  # (fp32 outputs end at 1e-38: the pair just made significant lies far below)
	assert not ok[5, 7] or want2[0][5, 7] < 1e-6
	assert p_close(changed[0][ok], want2[0][ok], 1e-6) and changed[0][5, 7] < 1e-6
	# a dense design has no lean form: the plan keeps calling the public function
	monkeypatch.setenv('NRM_DE_SPARSE', '0')
	dense = Single4Plan(d_x, d_y, dc, return_dot=False)
	dense.step()
	dense.step()
	assert dense.lean is False and p_close(dense.results()[0][ok], want2[0][ok], 1e-6)


def test_c_entry_single1_matches_the_plan_from_a_process_without_torch(tmp_path):
	"""nrm_association_tests_single1_host (numpy in / out, no torch in the process) runs the kernels of Single1Plan: the same results as the package gives
	in this process, for 5 covariates (statistics on the device) and for 12 (on the host, as in rounds 4-5)."""
	from normalisr_amd.single1 import association_tests_single1
	for nc in (5, 12):
		rng = np.random.default_rng(650 + nc)
		dx, dy, dc = _screen(rng, 30, 50, 5000, nc, valued=True)
		want = association_tests_single1(dx, dy, dc, return_dot=False)
		code = r'''
import sys, ctypes, numpy as np
sys.modules['torch'] = None
sys.path.insert(0, {root!r})
from normalisr_amd import _lib
_lib.prefer_host_entry(True)
lib = _lib.load()
rng = np.random.default_rng({seed})
nx, ny, n, nc = 30, 50, 5000, {nc}
dx = (rng.random((nx, n)) < 1.0 / nx).astype(np.float64)
dx[1] *= rng.uniform(0.5, 1.0, n)
dx[1, np.nonzero(dx[1])[0][0]] = 1.0
dy = rng.normal(size=(ny, n))
dy[:5] += 0.5 * dx[0]
dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
p, st, vy = (np.empty((nx, ny)) for _ in range(3))
vx = np.empty(nx)
rc = lib.nrm_association_tests_single1_host(dx.ctypes.data, 1, nx, dy.ctypes.data, 1, ny, dc.ctypes.data, 1, nc, n, 0, 0, p.ctypes.data, st.ctypes.data, None, vx.ctypes.data, vy.ctypes.data, 1)
assert rc == 0, lib.nrm_last_error()
np.savez({out!r}, p=p, st=st, vx=vx, vy=vy)
'''
		out = str(tmp_path / 's1_entry_{}.npz'.format(nc))
		r = subprocess.run([sys.executable, '-c', code.format(root=ROOT, seed=650 + nc, nc=nc, out=out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
		assert r.returncode == 0, r.stderr[-2000:]
		got = np.load(out)
		assert p_close(got['p'], want[0], 1e-10) and close(got['st'], want[1], 1e-10, 1e-13) and close(got['vx'], want[3], 1e-10) and close(got['vy'], want[4], 1e-10)


def test_host_entries_honour_the_selected_device(tmp_path):
	"""Round-5 advisory: the command line runs on the library's whole-problem entries, and nothing on that route looked at NORMALISR_DEVICE / device= --
	`NORMALISR_DEVICE=3 normalisr coex ...` ran on GPU 0.  The entries are bound to the selected GPU now (nrm_set_device: this thread, the entries' helper
	threads, the scratch pool); an index that is not there is the ValueError the torch engine raises for it."""
	rng = np.random.default_rng(660)
	ng, n = 60, 2500
	dt = np.log1p(rng.poisson(2.0, (ng, n))).astype(np.float64)
	dc = np.vstack([rng.normal(size=(1, n)), np.ones(n)])
	np.save(tmp_path / 'exp.npy', dt)
	np.save(tmp_path / 'cov.npy', dc)
	cmd = ['bash', os.path.join(ROOT, 'bin', 'normalisr'), 'coex', str(tmp_path / 'exp.npy'), str(tmp_path / 'cov.npy'), str(tmp_path / 'pv.npy')]
	env = {k: v for k, v in os.environ.items() if k not in ('NRM_HOST_ENTRY', 'NORMALISR_DEVICE')}
	r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, NORMALISR_DEVICE='0'))
	assert r.returncode == 0, r.stderr[-2000:]
	want = oracle.coex(dt, dc)
	assert p_close(np.load(tmp_path / 'pv.npy'), want[0])
	r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, NORMALISR_DEVICE='97'))
	assert r.returncode != 0 and 'GPU 97 requested' in r.stderr and 'ValueError' in r.stderr, r.stderr[-2000:]
	# in this process: the package's keyword on the host-entry route
	from normalisr_amd import _lib
	from normalisr_amd.association import association_tests
	prev = _lib.prefer_host_entry(True)
	try:
		with pytest.raises(ValueError, match='GPU 55 requested'):
			association_tests(dt, None, dc, device=55)
		got = association_tests(dt, None, dc, device=0)
		assert p_close(got[0], want[0])
	finally:
		_lib.prefer_host_entry(prev)


def test_single1_with_one_dimreduce_per_gene(golden):
	"""dimreduce as one value per row of dy for single=1 (association.py:374: the degrees of freedom of every (grouping, gene) pair lose their gene's value;
	rounds 3-5 raised NotImplementedError): each gene's P-values are those of the scalar call with its value, gamma and the variances are untouched --
	and that is what the oracle's restatement of the reference's own broadcast (a (n_y, 1) column, one tile) gives."""
	from normalisr_amd.single1 import association_tests_single1
	rng = np.random.default_rng(670)
	nx, ny, n, nc = 9, 23, 3000, 3
	dx, dy, dc = _screen(rng, nx, ny, n, nc)
	dr = rng.integers(0, 3, ny)
	dr[:3] = [0, 1, 2]
	got = association_tests_single1(dx, dy, dc, dimreduce=dr, return_dot=False)
	for v in range(3):
		want = oracle.association_tests(dx, dy, dc, single=1, dimreduce=v, return_dot=False)
		cols = dr == v
		assert p_close(got[0][:, cols], want[0][:, cols]) and close(got[1], want[1], 1e-9, 1e-12) and close(got[4], want[4], 1e-9)
	whole = oracle.association_tests(dx, dy, dc, single=1, dimreduce=dr.reshape(ny, 1), return_dot=False, bsx=nx, bsy=ny)
	assert p_close(got[0], whole[0])
	assert np.array_equal(association_tests_single1(dx, dy, dc, dimreduce=dr.reshape(ny, 1), return_dot=False)[0], got[0])
	with pytest.raises(ValueError):
		association_tests_single1(dx, dy, dc, dimreduce=dr[:-1])
	# the reference's own outputs for this case (G16, tests/golden/make_golden.py)
	g = golden('G16_single1_dimreduce')
	ref = association_tests_single1(g['dx'], g['dy'], g['dc'], dimreduce=g['dimreduce'], return_dot=False, lowmem=False)
	assert p_close(ref[0], g['p']) and close(ref[1], g['gamma'], 1e-9, 1e-12) and close(ref[2], g['alpha'], 1e-8, 1e-9) and close(ref[3], g['vx'], 1e-9) and close(ref[4], g['vy'], 1e-9)


def test_normvar_exp_against_numpy():
	"""The per-gene cell weights w_k^wt_g = exp(wt_g ln w_k) (norm.py:245) come from a table of 2^(j/64) and a degree-5 polynomial instead of the library's exp
	(both passes of normvar were bound by it): within 4 units of the last digit of numpy's exp -- 3e-16 of the exact value -- over the arguments weights
	produce and far beyond, exactly 1 at 0, 0 / inf where double precision ends."""
	import torch
	from normalisr_amd import _lib
	from normalisr_amd.engine import get_engine
	eng = get_engine()
	rng = np.random.default_rng(680)
	x = np.concatenate([rng.uniform(-6, 6, 400000), rng.uniform(-700, 700, 200000), -rng.exponential(1e-3, 1000), [0.0, 1e-300, -1e-300, 709.7, -745.2, -800.0, 720.0]])
	d_x = torch.from_numpy(x).cuda()
	d_o = torch.empty_like(d_x)
	_lib.check(eng.lib.nrm_normvar_exp_probe(d_x.data_ptr(), d_x.numel(), d_o.data_ptr(), eng._stream()))
	got, want = d_o.cpu().numpy(), np.exp(x)
	fin = np.isfinite(want) & (want > 1e-300)
	assert np.abs(got[fin] / want[fin] - 1).max() < 9e-16 and got[x == 0.0][0] == 1.0
	assert got[-2] == 0.0 and np.isinf(got[-1]) and not np.isnan(got).any()
	import mpmath as mp
	mp.mp.dps = 40
	idx = rng.integers(0, 600000, 2000)
	assert max(abs(mp.mpf(float(got[i])) / mp.exp(mp.mpf(float(x[i]))) - 1) for i in idx) < 3.5e-16


def test_normvar_host_entry_beyond_eight_covariates():
	"""nrm_normvar_host with 9 .. 32 covariates (round 5: NRM_E_UNSUPPORTED, so that `normalisr normvar` needed torch there): the Gram-launch form of norm.py in
	the library -- U = e^2, V = e^2 y, two launches of the fp64 Gram kernel, the genes' pseudo-inverses by the threaded Jacobi stack (integer ranks by inv_rank's
	rule: a covariate given twice), one pass for the result -- against the oracle (norm.py:166-289), and equal to the package's own route."""
	import normalisr_amd.normalisr as norm
	from normalisr_amd import _lib
	rng = np.random.default_rng(690)
	for nc, ng, n, dup in ((12, 70, 3000, False), (20, 40, 2500, True), (32, 25, 2100, False)):
		dt = rng.normal(size=(ng, n)) - 9
		dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
		if dup:
			dc[3] = dc[1]
		w, wt = np.exp(0.3 * rng.normal(size=n)), rng.uniform(0, 1.5, ng)
		wt[2] = 0.0
		ref = oracle.normvar(dt, dc, w, wt)
		pkg = norm.normvar(dt, dc, w, wt)
		prev = _lib.prefer_host_entry(True)
		try:
			got = norm.normvar(dt, dc, w, wt)
		finally:
			_lib.prefer_host_entry(prev)
		scale = np.abs(ref[0]).max()
		assert np.abs(got[0] - ref[0]).max() < 1e-9 * scale and close(got[1], ref[1], 1e-12, 1e-15), (nc, dup)
		assert np.abs(got[0] - pkg[0]).max() < 1e-11 * scale


_NO_TORCH_S4_REST = r"""
import sys
sys.modules['torch'] = None  # `import torch` raises ImportError from here on
sys.path.insert(0, sys.argv[1])
import numpy as np
import normalisr.normalisr as norm            # the drop-in import name
from normalisr_amd.association import association_tests
d = np.load(sys.argv[2])
dg, dt, dc, dr = d['dg'], d['dt'], d['dc'], d['dr']
dcr = np.vstack([dc, 2.0 * dc[:1]])  # a covariate given twice: A A^T is rank deficient, no closed form
out = {}
def put(name, res):
	for k, v in zip(('p', 'stat', 'alpha', 'vx', 'vy'), res):
		if v is not None:
			out[name + '_' + k] = v
put('rdef', association_tests(dg, dt, dcr, single=4, lowmem=False, return_dot=False))
put('mpc', association_tests(dg, dt, dc, single=4, lowmem=False, mpc=60))              # truncated pseudo-inverse
put('same', association_tests(dg, None, dc, single=4))                                   # every pair of groupings given all the others
put('samer', association_tests(dg, None, dcr, single=4, return_dot=False))
put('dr4', association_tests(dg, dt, dc, single=4, lowmem=False, dimreduce=dr))         # one dimreduce per gene, closed form per distinct value
put('dr4r', association_tests(dg, dt, dcr, single=4, dimreduce=dr))                     # ... and without a closed form
put('dr1', association_tests(d['dg1'], dt, dc, single=1, lowmem=False, dimreduce=dr.reshape(-1, 1)))
put('de', norm.de(dg, dt, dcr, single=4))
assert not any(m == 'torch' or m.startswith('torch.') for m, v in sys.modules.items() if v is not None)
np.savez(sys.argv[3], **out)
"""


def test_single4_without_a_closed_form_from_a_process_without_torch(tmp_path):
	"""The calls nrm_association_tests_single4_host answers NRM_E_UNSUPPORTED -- a rank-deficient A A^T, mpc, dy=None -- and one dimreduce per gene, in a
	process that cannot import torch: the package follows the reference's per-grouping algorithm (association.py:421-576) on Gram matrices from
	nrm_gram_host with P-values from nrm_pvalues_host, numpy and the library only.  Against the oracle's restatement of that algorithm."""
	import subprocess
	rng = np.random.default_rng(661)
	nx, ny, n = 24, 70, 3000
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dg = (rng.random((nx, n)) < 0.04).astype(np.float64)
	dcr = np.vstack([dc, 2.0 * dc[:1]])
	dg1 = (rng.random((nx, n)) < 1.0 / nx).astype(np.float64)
	dt = (rng.normal(size=(ny, n)) + 0.5 * dg[0] + 0.4 * dg[2] + 0.5 * dg1[1]).astype(np.float32)
	dr = rng.integers(0, 3, ny)
	np.savez(tmp_path / 'in.npz', dg=dg, dg1=dg1, dt=dt, dc=dc, dr=dr)
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	r = subprocess.run([sys.executable, '-c', _NO_TORCH_S4_REST, root, str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')], capture_output=True, text=True, timeout=900)
	assert r.returncode == 0, r.stderr[-3000:]
	o = np.load(tmp_path / 'out.npz')
	dt64 = dt.astype(np.float64)

	def same(name, ref, tol=1e-6):
		for k, v in zip(('p', 'stat', 'alpha', 'vx', 'vy'), ref):
			if v is None:
				assert name + '_' + k not in o.files
				continue
			got = o[name + '_' + k]
			assert got.shape == v.shape, (name, k)
			if k == 'p':
				assert p_close(got, v, 1e-6), (name, k)
			else:
				assert np.abs(got - v).max() <= tol * max(np.abs(v).max(), 1e-300), (name, k)

	same('rdef', oracle.association_tests(dg, dt64, dcr, single=4, lowmem=False, return_dot=False))
	same('mpc', oracle.association_tests(dg, dt64, dc, single=4, lowmem=False, mpc=60))
	same('same', oracle.association_tests(dg, None, dc, single=4))
	same('samer', oracle.association_tests(dg, None, dcr, single=4, return_dot=False))
	same('dr4', oracle.association_tests(dg, dt64, dc, single=4, lowmem=False, dimreduce=dr))
	same('dr4r', oracle.association_tests(dg, dt64, dcr, single=4, dimreduce=dr))
	same('dr1', oracle.association_tests(dg1, dt64, dc, single=1, lowmem=False, dimreduce=dr.reshape(-1, 1)))
	same('de', oracle.de(dg, dt64, dcr, single=4))
	assert o['rdef_p'].dtype == np.float32 and o['same_p'].dtype == np.float64  # the dtype of dy (of dx for dy=None), as the reference returns


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_binnet_on_wide_rows_of_every_density(dtype):
	"""k_binnet_rows on a 9000-wide matrix (the fixed tests stop at 700) whose rows are null P-values, moderately and very dense networks by turns -- with a long run
	of ties where the threshold falls, few distinct values, thousands of zeros, nothing below the cutoff: the histogram start and the counting passes at every density
	in one launch.  Booleans bit-exact against the oracle's restatement of binnet.py:77-173 on a sample of the rows (its loops are Python), the count the kernel
	returns against its own mask."""
	import torch
	from normalisr_amd import _lib
	from normalisr_amd.engine import get_engine
	eng = get_engine()
	rng = np.random.default_rng(677)
	ng = 9000
	p = rng.random((ng, ng))
	p **= np.array([1.0, 3.0, 8.0, 1.0, 2.0])[np.arange(ng) % 5][:, None]
	p[11, 100:4000] = p[11, 99]       # a long run of ties just where the threshold may fall
	p[12] = np.round(p[12], 2)        # few distinct values
	p[13, :5000] = 0.0                # zeros
	p[14] = 1.0                       # nothing below the cutoff
	pm = p.astype(dtype)
	d_p = torch.from_numpy(pm).cuda()
	code = _lib.NRM_F64 if dtype == np.float64 else _lib.NRM_F32
	rows = np.concatenate([np.arange(40), rng.choice(np.arange(40, ng), 260, replace=False)])
	for q in (0.05, 0.3):
		ref = np.zeros((len(rows), ng), dtype=bool)
		for k, i in enumerate(rows):
			off = np.arange(ng) != i
			ref[k, off] = oracle.bh(pm[i, off]) <= q
		out = torch.zeros((ng, ng), dtype=torch.uint8, device='cuda')
		total = torch.zeros(1, dtype=torch.int64, device='cuda')
		flags = torch.zeros(2, dtype=torch.int32, device='cuda')
		_lib.check(eng.lib.nrm_binnet(d_p.data_ptr(), code, ng, ng, q, out.data_ptr(), ng, total.data_ptr(), flags.data_ptr(), eng._stream()))
		got = out.cpu().numpy().astype(bool)
		assert int(flags[0].item()) == 0 and int(total.item()) == int(got.sum())
		assert np.array_equal(got[rows], ref)
		assert ref[2].sum() > 4096 and ref[13].sum() >= 5000 and ref[14].sum() == 0


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_normvar_plan_replays_the_public_call(dtype):
	"""NormvarPlan: the host's share of norm.normvar done once, a step = the three kernels on the same buffers (a HIP graph from the second step on) -- the bits of
	normvar(..., device_out=True), the oracle's per-gene loop (norm.py:232-259) within 1e-6 of the result's scale; the matrix rewritten in place between steps is
	what the next step normalises; the reference's assertion on a result that is not finite (norm.py:286) comes out of check()."""
	import torch
	import normalisr_amd.normalisr as norm
	from normalisr_amd.norm import NormvarPlan
	rng = np.random.default_rng(690)
	nt, ns, nc = 70, 3001, 4
	dt = (rng.normal(size=(nt, ns)) - 9).astype(dtype)
	dc = np.vstack([rng.normal(size=(nc - 1, ns)), np.ones((1, ns))])
	w, wt = np.exp(0.3 * rng.normal(size=ns)), rng.uniform(0, 1.5, nt)
	d_dt = torch.from_numpy(dt).cuda()
	pub = norm.normvar(d_dt, dc, w, wt, device_out=True)
	plan = NormvarPlan(d_dt, dc, w, wt)
	assert plan.lean
	for _ in range(4):
		out = plan.step()
	assert plan._graph.graph is not None and plan.check()
	assert torch.equal(out, pub[0]) and np.array_equal(plan.dcn, pub[1]) and out.dtype == pub[0].dtype
	ref = oracle.normvar(dt.astype(np.float64), dc, w, wt)
	got = plan.results()
	assert np.abs(got[0] - ref[0]).max() < 1e-6 * np.abs(ref[0]).max() and close(got[1], ref[1], 1e-12, 1e-15)
	d_dt[3] += 0.25 * torch.from_numpy(dc[0]).cuda().to(d_dt.dtype)  # in place: the next replay reads the new values
	out2 = plan.step()
	dt2 = d_dt.cpu().numpy()
	ref2 = oracle.normvar(dt2.astype(np.float64), dc, w, wt)
	assert np.abs(plan.results()[0] - ref2[0]).max() < 1e-6 * np.abs(ref2[0]).max() and out2 is out
	# a value that is not finite written into the resident matrix: the step runs (nothing of it is looked at on the host), check() raises the reference's
	# assertion (norm.py:286) -- and the counters start afresh
	d_dt[5, 17] = float('nan')
	plan.step()
	with pytest.raises(AssertionError):
		plan.check()
	d_dt[5, 17] = -9.0
	plan.step()
	assert plan.check()
	with pytest.raises(AssertionError):  # covariates that are all zero: 0 / 0 in every gene's pseudo-inverse, as in the reference (association.py:77-80 keeps s >= tol * 0)
		NormvarPlan(d_dt, np.zeros((2, ns)), w, wt)
