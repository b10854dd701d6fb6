"""Timing bounds of the GPU suite -- collected LAST (tests/conftest.py orders the tiers) so that no clock can hide a parity test behind `-x`.

Every bound is stated on a duration measured with HIP events around KERNELS on the launch stream (the `kernels_ms` / `roofline.kernel_ms`
figures of bench.py), never on the host's wall clock of a 2-step sample, and carries at least 5x slack over the figure measured on the
boxes of rounds 4-6 (profiles/r0*_bench_default.json): it fails on a kernel that lost a factor, not on a box whose host or clock is slow.
Step times (which include whatever the host does between launches) are printed, and bounded only relative to the step's own kernels with
the same slack."""
import pytest

pytestmark = pytest.mark.gpu

SLACK = 5.0
# kernel, measured ms (rounds 4-6, three boxes)
MEASURED = {
	('coex_c2', 'gram'): 2.15, ('coex_c2', 'residualize'): 0.20, ('coex_c2', 'sweep'): 0.16,
	('coex_c5', 'gram'): 62.5, ('coex_c5', 'residualize'): 10.8,
	('coex_c2_f64', 'gram'): 4.0,
	('de_c3', 'gram'): 1.87,
	('de_c4', 'de_sparse'): 1.85,
	('de_c4_single4', 'de_sparse'): 2.0,
	('de_c4_single1', 's1_stream'): 1.31, ('de_c4_single1', 's1_cells'): 0.35,
}


def test_kernel_times_have_not_lost_a_factor(bench_default):
	ex = bench_default['_detail']
	report, bad = [], []
	for (w, k), ms in sorted(MEASURED.items()):
		got = ex[w]['kernels_ms'].get(k)
		assert got is not None and got > 0, (w, k, ex[w]['kernels_ms'])
		report.append('{}.{}: {:.3f} ms (measured {:.2f})'.format(w, k, got, ms))
		if got > SLACK * ms:
			bad.append(report[-1])
	print('\n'.join(report))
	assert not bad, bad


def test_rooflines_are_fractions(bench_default):
	"""`achieved` never above the peak of the unit that executes (a fraction over 1 means the algorithmic work or the clock is wrong)."""
	for w, v in bench_default['_detail'].items():
		assert 0 < v['roofline']['frac'] < 1.2, (w, v['roofline'])


def test_steps_cost_about_their_kernels(bench_default):
	"""A resident step is its kernels: the host between the launches (read-backs, small factorisations, allocations) may not multiply it.
	Relative, with the same slack; printed either way -- round 5's driver box took 18.4 ms for the single=1 step whose kernels take 1.7."""
	ex = bench_default['_detail']
	report, bad = [], []
	for w in ('coex_c2', 'de_c3', 'de_c4', 'de_c4_single4', 'de_c4_single1', 'coex_c5', 'normvar_c2'):
		kern = sum(v for v in ex[w]['kernels_ms'].values() if v)
		report.append('{}: step {:.3f} ms, its kernels {:.3f} ms'.format(w, ex[w]['ms_per_step'], kern))
		if ex[w]['ms_per_step'] > SLACK * kern + 1.0:
			bad.append(report[-1])
	print('\n'.join(report))
	assert not bad, bad
