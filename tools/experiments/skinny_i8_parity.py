"""The parity test of the streaming de pass on the int8 matrix cores (tools/experiments/nrm_skinny_i8.hip), as it ran in tests/test_gpu_round3.py
while the kernel was part of the shipped library (rounds 3-4, opt-in NRM_DE_I8=1).  The kernel was correct and slower than the fp64 streaming
kernel on every BASELINE shape (2.48 against 1.87 ms on configs[2]) and left the library in round 5 with its entry
points (nrm_skinny_i8, nrm_row_scales, the integer arguments of nrm_de_small_sweep).  Kept for whoever picks the formulation up again: it needs
those entry points restored (git show 91ead47:normalisr_amd/csrc/nrm_sweep.hip, :include/normalisr_hip.h, :normalisr_amd/engine.py)."""
import numpy as np
import pytest

# archived with its kernel: the entry points it drives are no longer exported by the library (see the header); not collected (the file name
# does not match test_*.py, and the repo-root conftest.py ignores tools/)
pytestmark = pytest.mark.skip(reason="archived experiment: needs the entry points of the kernel restored")

@pytest.mark.parametrize('dtype,n,ny,nc', [(np.float32, 4096 + 48, 1000, 20), (np.float64, 8192, 530, 3), (np.float32, 20000, 700, 30)])
def test_streaming_de_on_the_int8_matrix_cores(eng, dtype, n, ny, nc, monkeypatch):
	"""A resident DePlan on the streaming path (n_x + n_cov <= 31): the first step runs the fp64 kernel and takes the fixed-point scale
	of every expression row, the following steps stream the raw rows through csrc/nrm_skinny_i8.hip -- cut into 46-bit digits on the
	fly, contracted exactly on the int8 matrix cores, corrected and certified by the sweep -- and must give the first step's and
	the oracle's answers (association.py:224-249); rows changed in place are detected and the step redone on the fp64 kernel."""
	import torch
	from normalisr_amd.distributed import DePlan
	monkeypatch.setenv('NRM_GRAPH', '0')
	monkeypatch.setenv('NRM_DE_I8', '1')  # (opt-in: the kernel is correct but not yet faster than the fp64 one, see its header)
	rng = np.random.default_rng(91)
	dx = (rng.random((1, n)) < 0.4).astype(dtype)
	lat = rng.normal(size=n)
	dy = (2.0 + rng.normal(size=(ny, n)) * rng.uniform(0.2, 2, (ny, 1)) + 0.3 * dx + 0.2 * rng.normal(size=(ny, 1)) * lat).astype(dtype)
	dy[3] = np.log1p(rng.poisson(0.01, n))  # a sparse row, a constant row and a row with one spike
	dy[4] = 1.5
	dy[5] = 1e-3 * rng.normal(size=n)
	dy[5, 77] = 30.0
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]).astype(dtype)
	ty = torch.from_numpy(dy).cuda()
	plan = DePlan(torch.from_numpy(dx).cuda(), ty, dc)
	assert plan.streaming()
	eng.trace = []
	plan.step()
	first = plan.results()
	assert 'yscale' in plan._state and any(t[0] == 'row scales' for t in eng.trace)
	plan.step()
	plan.step()
	eng.trace = None
	again = plan.results()
	guard = dict(eng.last_guard)
	assert not guard['fallback'] and guard['hits'] == 0 and 0 < guard['worst'] < eng.guard_tol, guard
	po = oracle.de(dx.astype(np.float64), dy.astype(np.float64), dc.astype(np.float64))
	tiny = 2.3e-308 if dtype == np.float64 else 1e-37  # (below: subnormal or 0 in the output type)
	ok = po[0] >= tiny
	ok[0, 4] = False  # the constant gene: its residuals are cancellation noise, in the reference as here -- P is near 1 either way
	tol = 1e-6 if dtype == np.float64 else 2e-5  # (fp32 results)
	for res in (first, again):
		assert relerr(res[0][ok], po[0][ok]) < tol and close(np.delete(res[3], 4), np.delete(po[4][0], 4), 1e-6 if dtype == np.float64 else 1e-5, 1e-12)
		assert res[0][0, 4] > 0.9 and po[0][0, 4] > 0.9
	assert relerr(again[0][ok], first[0][ok]) < (1e-7 if dtype == np.float64 else 2e-6) and close(again[1], first[1], 1e-6, 1e-9)
	# rows changed in place under the plan: the scales no longer fit -> noticed by the sweep, step redone on the fp64 kernel
	ty[10] *= 1000.0
	ty[11] = 0.0
	plan.step()
	changed = plan.results()
	assert eng.last_guard['fallback']
	dy2 = ty.cpu().numpy()
	po2 = oracle.de(dx.astype(np.float64), dy2.astype(np.float64), dc.astype(np.float64))
	ok2 = po2[0] >= tiny
	ok2[0, 4] = False
	assert relerr(changed[0][ok2], po2[0][ok2]) < tol
