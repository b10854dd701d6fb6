"""GPU parity tests added in round 2: the BASELINE configs at their full cell counts (configs[2] exactly, configs[3]
exactly and sharded, the per-rank shape of configs[4] in fp64 at 500 000 cells), alpha with return_dot, allocator
poisoning of single=4, the unpinned-copy fallback, the self-launching bench, and the row-block binnet.  Same parity
bar as test_gpu_parity.py: integers/zeros/shapes bit-exact, r, t and p within 1e-6 relative (floors stated)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from conftest import ROOT, relerr
from test_gpu_parity import close, p_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def norm():
	import normalisr_amd.normalisr as norm
	return norm


@pytest.fixture(scope='module')
def eng():
	from normalisr_amd.engine import get_engine
	return get_engine()


@pytest.mark.parametrize('path', ['auto', 'general'])
def test_alpha_with_return_dot(path, monkeypatch):
	"""association_tests(lowmem=False, return_dot=True): the reference computes alpha from gamma inside association_test_1
	(association.py:238-243) whatever return_dot says; return_dot only rescales the statistic afterwards (:1044-1048)."""
	from normalisr_amd.association import association_tests
	monkeypatch.setenv('NRM_DE_PATH', path)
	rng = np.random.default_rng(909)
	n, nc = 900, 3
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
	dx = (rng.random((4, n)) < 0.4).astype(np.float64)
	dy = rng.normal(size=(150, n)) + 0.7 * dc[0] + 0.3 * dx[1]
	for return_dot in (True, False):
		p, st, al, vx, vy = association_tests(dx, dy, dc, lowmem=False, return_dot=return_dot)
		po, so, ao, vxo, vyo = oracle.association_tests(dx, dy, dc, lowmem=False, return_dot=return_dot)
		assert al.shape == (4, 150, nc) and np.abs(ao).max() > 0.1
		assert p_close(p, po) and close(st, so, floor=1e-12) and close(al, ao, floor=1e-9)
	# fp32 in: the conversion of the rounded covariance back to gamma stays within the tolerance
	p, st, al, vx, vy = association_tests(dx.astype(np.float32), dy.astype(np.float32), dc.astype(np.float32), lowmem=False, return_dot=True)
	po, so, ao, vxo, vyo = oracle.association_tests(dx, dy.astype(np.float32).astype(np.float64), dc.astype(np.float32).astype(np.float64), lowmem=False, return_dot=True)
	assert al.dtype == np.float32 and close(al, ao, 1e-5, 1e-4)


def test_host_entry_alpha_with_return_dot(eng):
	"""The C host entry (no torch) returns alpha for return_dot=1 instead of failing after the whole computation."""
	import ctypes
	from normalisr_amd import _lib
	from normalisr_amd.association import inv_rank
	rng = np.random.default_rng(910)
	n, nc, nx, ny = 500, 2, 3, 40
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	dx = (rng.random((nx, n)) < 0.4).astype(np.float64)
	dy = rng.normal(size=(ny, n)) + 0.5 * dc[0]
	dci, rank = inv_rank(dc @ dc.T)
	p, st, al = np.empty((nx, ny)), np.empty((nx, ny)), np.empty((nx, ny, nc))
	vx, vy = np.empty(nx), np.empty(ny)
	ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
	rc = eng.lib.nrm_association_tests_host(ptr(dx), _lib.NRM_F64, nx, ptr(dy), _lib.NRM_F64, ny, ptr(dc), _lib.NRM_F64, nc, n, ptr(dci), rank, 0, 1,
											ptr(p), ptr(st), ptr(al), ptr(vx), ptr(vy), None, None, _lib.NRM_F64)
	assert rc == 0, eng.lib.nrm_last_error()
	po, so, ao, vxo, vyo = oracle.association_tests(dx, dy, dc, lowmem=False, return_dot=True)
	assert p_close(p, po) and close(st, so, floor=1e-12) and close(al, ao, floor=1e-9)


def test_single4_after_poisoned_allocator(norm):
	"""single=4 with m = nx + nc not a multiple of 128 right after the caching allocator held NaNs: the Y A^T buffer is the
	K-operand of the next contraction, so its padding columns must be zeros, not whatever the allocator hands back."""
	import torch
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(911)
	n, nx, ny, nc = 1200, 21, 300, 2
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	dx = (rng.random((nx, n)) < 0.1).astype(np.float64)
	dy = rng.normal(size=(ny, n)) + 0.4 * dx[3]
	ref = oracle.association_tests(dx, dy, dc, single=4, return_dot=False)
	for _ in range(3):
		junk = [torch.full((384, 128 * k), float('nan'), dtype=torch.float64, device='cuda') for k in (1, 2, 3)]
		del junk
		got = association_tests(dx, dy, dc, single=4, return_dot=False)
		assert p_close(got[0], ref[0]) and close(got[1], ref[1], floor=1e-12) and close(got[4], ref[4], 1e-9)


def test_pin_failure_falls_back_and_leaves_no_sticky_error(eng, monkeypatch):
	"""A failed HIP call reported through the C ABI (here: unpinning a range that was never page-locked) must not surface
	again as the next launch's error; and Engine.download / download_into fall back to a pageable copy when the result
	array cannot be page-locked (hipHostRegister is made to fail: on this ROCm even a double registration succeeds)."""
	import torch
	from normalisr_amd import _lib
	t = torch.arange(1 << 19, dtype=torch.float64, device='cuda')  # 4 MB: above the pinning threshold
	out = np.empty(1 << 19)
	with pytest.raises(RuntimeError):
		eng.host_unpin(out)  # hipHostUnregister of an unregistered range fails ...
	d_r2 = torch.full((1000, ), 0.01, dtype=torch.float64, device='cuda')
	d_p = torch.empty_like(d_r2)
	_lib.check(eng.lib.nrm_pvalues_from_r2(d_r2.data_ptr(), 1000, 50.0, d_p.data_ptr(), eng._stream()))  # ... and the next launch check is clean
	torch.cuda.synchronize()
	assert np.array_equal(eng.download_into(t, out), np.arange(1 << 19, dtype=np.float64))  # page-locked copy

	def no_pin(a):
		raise RuntimeError('hipHostRegister failed: out of memory (simulated locked-memory limit)')
	monkeypatch.setattr(eng, 'host_pin', no_pin)
	monkeypatch.setattr(eng.pool, 'limit', 0)  # no recycled page-locked blocks either: fresh numpy memory that cannot be locked
	out[:] = 0
	assert np.array_equal(eng.download_into(t, out), np.arange(1 << 19, dtype=np.float64))  # pageable fallback
	assert np.array_equal(eng.download(t), np.arange(1 << 19, dtype=np.float64))


def test_many_covariates_beyond_one_lds_table(norm):
	"""300 covariates (the first K1 held its OLS tables in a fixed 256-entry LDS array): dynamic LDS tables, same parity bar."""
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(912)
	n = 2000
	dy = rng.normal(size=(50, n))
	dx = (rng.random((3, n)) < 0.3).astype(np.float64)
	for nc in (300, 1000):  # 1000: tables beyond the default 48 KB dynamic-LDS window
		dc = rng.normal(size=(nc, n))
		p, g, a, vx, vy = association_tests(dx, dy, dc, return_dot=False)
		po, go, ao, vxo, vyo = oracle.association_tests(dx, dy, dc, return_dot=False)
		assert p_close(p, po) and close(g, go, floor=1e-12) and close(vy, vyo, 1e-9)
	with pytest.raises(ValueError):
		association_tests(dx, dy[:, :1500], rng.normal(size=(1400, 1500)), return_dot=False)  # n <= rank + 1 (association.py:213-216)


def test_binnet_row_blocks_equal_whole(eng):
	"""nrm_binnet_rows on row blocks (what a rank of a sharded coex runs on its rows) == nrm_binnet on the whole matrix ==
	the oracle, bit for bit, for fp32 and fp64 matrices."""
	import torch
	from normalisr_amd import _lib
	rng = np.random.default_rng(913)
	ng = 700
	p = rng.random((ng, ng))**3
	p = np.triu(p, 1) + np.triu(p, 1).T
	p[5, 7:60] = p[5, 6]  # ties
	for dt in (np.float64, np.float32):
		pm = p.astype(dt)
		ref = oracle.binnet(pm, 0.2)
		d_p = torch.from_numpy(pm).cuda()
		out = torch.zeros((ng, ng), dtype=torch.uint8, device='cuda')
		tot = 0
		for a, b in ((0, 128), (128, 333), (333, ng)):
			total = torch.zeros(1, dtype=torch.int64, device='cuda')
			flags = torch.zeros(2, dtype=torch.int32, device='cuda')
			blk = d_p[a:b].contiguous()
			_lib.check(eng.lib.nrm_binnet_rows(blk.data_ptr(), _lib.NRM_F64 if dt == np.float64 else _lib.NRM_F32, b - a, ng, ng, a, 0.2,
											   out[a:b].data_ptr(), ng, total.data_ptr(), flags.data_ptr(), eng._stream()))
			tot += int(total.item())
			assert int(flags[0].item()) == 0
		assert np.array_equal(out.cpu().numpy().astype(bool), ref) and tot == int(ref.sum())


def test_config2_exact_shape_sampled(eng):
	"""BASELINE configs[2] at its exact shape: norm.de, 1 grouping x 20 000 genes x 100 000 cells, 20 covariates, fp32,
	the expression matrix generated in HBM (8 GB); through DePlan (what bench.py --workload de_c3 times, streaming kernel)
	and through the general path; sampled genes against the oracle on the same rows."""
	import torch
	from normalisr_amd.distributed import DePlan
	ny, n, nc = 20000, 100000, 20
	gen = torch.Generator(device='cuda').manual_seed(31)
	dc = torch.cat([torch.randn((nc - 1, n), generator=gen, device='cuda'), torch.ones((1, n), device='cuda')])
	dx = (torch.rand((1, n), generator=gen, device='cuda') < 0.5).float()
	dy = torch.randn((ny, n), generator=gen, device='cuda') * 1.5 + 3.0
	eff = torch.linspace(0.0, 0.3, 64, device='cuda')
	dy[:64] += eff[:, None] * dx[0]
	rows = np.concatenate([np.arange(64), np.arange(9990, 10010), np.arange(ny - 20, ny)])
	sub = dy[torch.from_numpy(rows).cuda()].cpu().numpy().astype(np.float64)
	po, go, ao, vgo, vto = oracle.de(dx.cpu().numpy().astype(np.float64), sub, dc.cpu().numpy().astype(np.float64))
	assert po.min() < 1e-30 and po.max() > 0.3
	plan = DePlan(dx, dy, dc)
	assert plan.streaming()
	plan.step()
	p, g, vg, vt = plan.results()
	assert p.shape == (1, ny) and p.dtype == np.float32
	assert close(p[:, rows], po, 1e-6, 1e-38) and close(g[:, rows], go, 1e-6, 1e-7) and close(vt[rows], vto[0], 1e-6) and close(vg, vgo, 1e-6)
	os.environ['NRM_DE_PATH'] = 'general'
	try:
		plan = DePlan(dx, dy, dc)
		assert not plan.streaming()
		plan.step()
		p2, g2, vg2, vt2 = plan.results()
	finally:
		del os.environ['NRM_DE_PATH']
	assert close(p2[:, rows], po, 1e-6, 1e-38) and close(g2[:, rows], go, 1e-6, 1e-7) and close(vt2[rows], vto[0], 1e-6)
	# the two paths agree on ALL 20 000 genes (fp32 outputs: to the last bit or two)
	assert close(p2, p, 1e-5, 1e-38) and close(g2, g, 1e-5, 1e-7)


def test_config3_exact_shape_sharded(eng):
	"""BASELINE configs[3] at its exact shape: 1 000 gRNAs x 15 000 genes x 50 000 cells fp32, gene rows sharded over 2 and 8
	DePlan ranks (run one after the other on this GPU: there is no collective on the de data path); sampled genes vs oracle,
	and the 8-way shards equal the 2-way shards."""
	import torch
	from normalisr_amd.distributed import DePlan
	nx, ny, n, nc = 1000, 15000, 50000, 5
	gen = torch.Generator(device='cuda').manual_seed(41)
	dc = torch.cat([torch.randn((nc - 1, n), generator=gen, device='cuda'), torch.ones((1, n), device='cuda')])
	dx = (torch.rand((nx, n), generator=gen, device='cuda') < 0.01).float()
	dy = torch.randn((ny, n), generator=gen, device='cuda')
	dy[:15] += 0.5 * dx[:15]
	rows = np.concatenate([np.arange(24), np.arange(7490, 7510), np.arange(ny - 20, ny)])
	sub = dy[torch.from_numpy(rows).cuda()].cpu().numpy().astype(np.float64)
	po, go, ao, vxo, vyo = oracle.association_tests(dx.cpu().numpy().astype(np.float64), sub, dc.cpu().numpy().astype(np.float64), return_dot=False)
	assert po.min() < 1e-20
	res = {}
	for world in (2, 8):
		R = ny // world
		parts = []
		for rank in range(world):
			plan = DePlan(dx, dy[rank * R:(rank + 1) * R], dc, rank=rank, world=world)
			assert not plan.streaming()
			plan.step()
			parts.append(plan.results())
		res[world] = (np.concatenate([q[0] for q in parts], axis=1), np.concatenate([q[1] for q in parts], axis=1),
					  np.concatenate([q[3] for q in parts]))
		p, g, vt = res[world]
		assert p.shape == (nx, ny)
		assert close(p[:, rows], po, 1e-6, 1e-38) and close(g[:, rows], go, 1e-6, 1e-7) and close(vt[rows], vyo, 1e-6)
		assert close(parts[0][2], vxo, 1e-6)
	assert close(res[8][0], res[2][0], 1e-5, 1e-38) and close(res[8][1], res[2][1], 1e-5, 1e-7)


def test_config4_rank_shape_fp64_500k_cells(eng):
	"""BASELINE configs[4], the shape one of 8 ranks works on: 3 840 gene rows (two blocks of 1 920) x 500 000 cells in fp64,
	generated in HBM (15 GB), coex through the engine with Pearson r and t; sampled rows against the oracle at 1e-6 on
	p, r and t (dof = 499 996: p spans 1 ... 1e-200 for |r| up to 0.05), exact zero diagonal and symmetry."""
	import torch
	from normalisr_amd.association import inv_rank
	ng, n, nc = 3840, 500000, 3
	gen = torch.Generator(device='cuda').manual_seed(51)
	lat = torch.randn((1, n), generator=gen, device='cuda', dtype=torch.float64)
	load = torch.randn((ng, 1), generator=gen, device='cuda', dtype=torch.float64)
	dt = torch.randn((ng, n), generator=gen, device='cuda', dtype=torch.float64)
	dt.addcmul_(load, lat, value=0.06)
	rng = np.random.default_rng(52)
	dc = np.vstack([rng.standard_normal((nc - 1, n)), np.ones((1, n))])
	dci, rank = inv_rank(dc @ dc.T)
	rows = np.concatenate([np.arange(16), np.arange(1912, 1928), np.arange(ng - 16, ng)])
	idx = torch.from_numpy(rows).cuda()
	sub = dt[idx].cpu().numpy()
	res = eng.association_single0(dt, None, dc, dci, rank, 0, True, False, np.float64, want_rt=True)
	po, do, vo = oracle.coex(sub, dc)
	dof = n - 1 - rank
	ro, to = oracle.pearson_r_t(do, vo, vo, dof)
	sel = np.ix_(rows, rows)
	off = ~np.eye(len(rows), dtype=bool)
	assert po[off].min() < 1e-40 and po[off].max() > 0.5
	assert p_close(res['p'][sel], po) and close(res['stat'][sel], do, floor=1e-13) and close(res['vary'][rows], vo, 1e-10)
	assert close(res['r'][sel][off], ro[off], floor=1e-12) and close(res['t'][sel][off], to[off], floor=1e-6)
	assert (np.diag(res['p']) == 0).all() and (res['p'] == res['p'].T).all() and (res['stat'] == res['stat'].T).all()


from conftest import run_bench as _run_bench  # noqa: E402


def test_bench_self_launches_two_ranks_on_one_gpu():
	"""`python bench.py --gpus 2` as a plain invocation starts its own ranks (here both on this GPU over gloo: functional
	check of the launcher and the N>1 path; RCCL needs one GPU per rank); the contract line is the last line of stdout."""
	out = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'coex_c2', '--genes', '1200', '--cells', '2000', '--no-extras'],
					 dict(NRM_DIST_BACKEND='gloo', NRM_SHARE_GPU='1'))
	assert out['n_gpus'] == 2 and out['ranks_seen_by_collective'] == 2 and out['value'] > 0
	assert out['roofline']['bound'] == 'mfma' and out['roofline']['kernel_ms'] > 0 and 'exchange' in out['kernels_ms']
	# the line checks itself: >= 12 pooled gene rows, their N-rank P-values against this device's own one-rank values, 1e-6
	sc = out['self_check']
	assert sc['ok'] and sc['ranks'] == 2 and sc['ranks_seen_by_collective'] == 2 and sc['gene_rows_checked'] >= 12 and sc['max_relative_p_difference'] <= 1e-6
	assert out['config']['exchange'] == 'all-gather of raw fp32 blocks'  # below 2048 cells: fp64 engine, raw rows travel
	out = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'coex_c2', '--genes', '1200', '--cells', '8192', '--no-extras'],
					 dict(NRM_DIST_BACKEND='gloo', NRM_SHARE_GPU='1'))
	assert out['n_gpus'] == 2 and out['value'] > 0 and out['config']['exchange'] == 'all-gather of raw fp32 blocks'  # auto on 2 ranks
	out = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'coex_c2', '--genes', '1200', '--cells', '8192', '--no-extras'],
					 dict(NRM_DIST_BACKEND='gloo', NRM_SHARE_GPU='1', NRM_EXCHANGE='chunks'))
	assert out['n_gpus'] == 2 and out['value'] > 0 and 'in 2 cell chunks' in out['config']['exchange']
	assert out['kernels_ms']['gram'] > 0 and out['kernels_ms']['exchange'] >= 0
	# the N > 1 DEFAULT is the workload the 8-GPU target is quoted on: configs[4] (here at a reduced per-rank block so that two ranks
	# share one GPU in seconds): fp64 rows, digit planes exchanged in cell chunks, exchange accounting in the line -- and the SAME key,
	# `scaling_series`, that the N = 1 line carries for this workload
	out = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--c5-rows', '384', '--c5-cells', '40000', '--no-extras'],
					 dict(NRM_DIST_BACKEND='gloo', NRM_SHARE_GPU='1'))
	assert out['n_gpus'] == 2 and out['ranks_seen_by_collective'] == 2 and 'BASELINE configs[4]' in out['config']['workload']
	assert out['config']['genes'] == 768 and out['config']['cells'] == 40000 and out['dtype'].startswith('i8 digits')
	assert 'cell chunks' in out['config']['exchange'] and out['config']['exchange_bytes_per_rank'] == 384 * 40000 * 6
	assert out['kernels_ms']['exchange'] >= 0 and out['kernels_ms']['gram'] > 0 and out['guard']['uncertified_pairs'] == 0
	assert out['self_check']['ok'] and out['self_check']['gene_rows_checked'] >= 12 and out['self_check']['max_relative_p_difference'] <= 1e-6
	# ... and fails, non-zero, when a rank's P-values are not what one rank computes (one value of the last rank moved by 1e-3)
	import subprocess
	r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--c5-rows', '384', '--c5-cells', '40000', '--no-extras'],
					   env=dict(os.environ, NRM_DIST_BACKEND='gloo', NRM_SHARE_GPU='1', NRM_BENCH_FAULT='p'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
	assert r.returncode != 0 and 'failed its own check' in r.stderr, (r.returncode, r.stderr[-1500:])
	ser = out['scaling_series']
	assert ser['ranks'] == 2 and ser['value'] == out['value'] and ser['ms_per_step'] == out['ms_per_step'] and ser['workload'].startswith('coex_c5') and 'scaling_series' in out['config']
	one = _run_bench(['--steps', '2', '--warmup', '1', '--c5-rows', '384', '--c5-cells', '40000', '--cpu-seconds', '0', '--e2e', '0', '--extras', 'coex_c5', '--extras-steps', '2'], {})
	assert one['n_gpus'] == 1 and one['config']['genes'] == 5000 and one['scaling_series']['ranks'] == 1 and one['scaling_series']['workload'] == ser['workload']
	assert one['scaling_series']['ms_per_step'] == one['_detail']['coex_c5']['ms_per_step'] and one['scaling_series']['tests_per_step'] == 384 * 383 // 2


def test_bench_default_line_carries_the_other_configs(bench_default):
	"""The default N=1 run: configs[1] as `value` on a contract line under 4 KB that is the LAST line of stdout, with one summary row per
	extra workload (de_c3 / de_c4 / coex_c5 = the per-rank slice / configs[4] whole on this one GPU / ...) and the scaling series; the full
	record of every workload -- roofline, kernel split, the guard's verdict -- on the lines before it.  CONTENT only: which kernels ran, which
	keys are there, what the guard said.  No bound on a time here (round 5's driver run lost 101 tests behind `ms_per_step < 12` on a box
	whose host was slow); the timing bounds are tests/test_zz_perf_gpu.py, collected last."""
	out = bench_default
	assert out['n_gpus'] == 1 and out['config']['genes'] == 5000 and out['dtype'].startswith('i8 digits') and out['roofline']['kernel'] == 'k_gram_i8'
	ex = out['_detail']
	names = {'de_c3', 'de_c4', 'de_c4_single4', 'de_c4_single1', 'coex_c5', 'coex_c5_f64', 'coex_c2_f64', 'binnet_c5', 'normvar_c2', 'chain_c2', 'coex_c5_full_1gpu'}
	assert set(ex) == names | {'coex_c2'} and set(out['extra_workloads']) == names, (sorted(ex), sorted(out['extra_workloads']))
	for w in names:  # the summary rows are the records' numbers
		row, rec = out['extra_workloads'][w], ex[w]
		assert abs(row['value'] / rec['value'] - 1) < 1e-3 and abs(row['ms'] - rec['ms_per_step']) < 1e-3 and abs(row['frac'] - rec['roofline']['frac']) < 1e-3, (w, row)
	# (b) one workload at every N: the per-rank slice of configs[4], here from the coex_c5 extra
	ser = out['scaling_series']
	assert ser['ranks'] == 1 and ser['value'] == ex['coex_c5']['value'] and ser['ms_per_step'] == ex['coex_c5']['ms_per_step'] and ser['tests_per_step'] == 3750 * 3749 // 2
	assert ex['de_c4_single1']['roofline']['kernel'] == 'k_s1_stream'
	assert ex['coex_c2_f64']['roofline']['kernel'] == 'k_gram_f64' and ex['coex_c2_f64']['dtype'] == 'f64' and ex['coex_c2_f64']['roofline']['traffic'] is None
	assert 'traffic_source' not in ex['coex_c2_f64']['roofline']
	# configs[4]'s literal-fp64 companion: the per-rank slice on the fp64 matrix cores, priced against their peak
	assert ex['coex_c5_f64']['roofline']['kernel'] == 'k_gram_f64' and ex['coex_c5_f64']['dtype'] == 'f64' and ex['coex_c5_f64']['roofline']['peak'] == 78.6
	assert ex['coex_c5_f64']['config']['tests_per_step'] == ex['coex_c5']['config']['tests_per_step']
	assert ex['de_c4_single4']['roofline']['kernel'] == 'k_de_sparse' and ex['de_c4_single4']['guard']['uncertified_pairs'] == 0 and not ex['de_c4_single4']['guard']['fp64_rerun']
	assert out['guard']['uncertified_pairs'] == 0 and 0 < out['guard']['largest_relative_p_error_bound'] < out['guard']['tolerance']
	assert ex['coex_c5_full_1gpu']['config']['tests_per_step'] == 30000 * 29999 // 2 and ex['coex_c5_full_1gpu']['guard']['uncertified_pairs'] == 0
	# configs[3]'s design is gRNA incidence (1 % of the entries set): the sparse-design kernels, not K1 + K2 (tests/test_gpu_round4.py holds both to the oracle)
	assert ex['de_c4']['roofline']['kernel'] == 'k_de_sparse' and ex['de_c4']['kernels_ms']['de_sparse'] > 0
	# (a) its traffic is that of the kernels the step ran (or null) -- never the dense path's k_gram_i8
	roof = ex['de_c4']['roofline']
	assert roof['traffic'] is None or ('k_de_sparse' in roof['traffic_source'] and 'k_gram_i8' not in roof['traffic_source'] and roof['traffic'] < 4 * roof['algorithmic_bytes'])
	# a cold call (a design tensor the engine has not seen: lists built inside the call) is reported beside the resident step
	assert ex['de_c4']['cold_ms'] > 0 and out['extra_workloads']['de_c4']['cold_ms'] == round(ex['de_c4']['cold_ms'], 3)
	assert ex['chain_c2']['config']['edges_kept'] > 0
	assert ex['de_c3']['roofline']['bound'] == 'hbm' and ex['de_c4']['roofline']['bound'] == 'hbm' and ex['coex_c5']['roofline']['bound'] == 'mfma'
	for k, v in ex.items():
		assert 'error' not in v, (k, v)
		assert v['value'] > 0 and v['ms_per_step'] > 0 and v['roofline']['frac'] > 0, (k, v['roofline'])


def test_single4_variants_golden_and_oracle(golden, norm):
	"""single=4 to the reference's contract (association.py:421-576,926-980): dy=None (closed form from the inverse Gram
	matrix), one dimreduce per gene, mpc-truncated pseudo-inverses and rank-deficient covariates (the reference's
	per-grouping algorithm on device-computed Gram matrices) -- golden G10 and a larger seeded case against the oracle."""
	from normalisr_amd.association import association_tests
	g = golden('G10_single4')
	dg, dc, dt = g['dg'], g['dc'], g['dt']
	for rd in (1, 0):
		p, d, a, vx, vy = association_tests(dt[:14], None, dc, single=4, return_dot=bool(rd))
		assert a is None and vx is None and p.shape == (14, 14)
		assert p_close(p, g['sx_p_rd%d' % rd]) and close(d, g['sx_dot_rd%d' % rd], floor=1e-12) and close(vy, g['sx_vy_rd%d' % rd], 1e-9)
		assert (np.diag(p) == 0).all() and (p == p.T).all() and (d == d.T).all() and (np.diag(vy) == 1).all()
	p, gam, a, vg, vt = norm.de(dg, dt, dc, single=4, dimreduce=g['dr'])
	assert p_close(p, g['dr_p']) and close(gam, g['dr_gamma'], floor=1e-12) and close(vg, g['dr_varg'], 1e-9) and close(vt, g['dr_vart'], 1e-9)
	p, gam, a, vg, vt = norm.de(dg, dt, dc, single=4, mpc=5, lowmem=False)
	assert p_close(p, g['mpc_p']) and close(gam, g['mpc_gamma'], floor=1e-12) and close(a, g['mpc_alpha'], floor=1e-9)
	assert close(vg, g['mpc_varg'], 1e-9) and close(vt, g['mpc_vart'], 1e-9)
	p, gam, a, vg, vt = norm.de(dg, dt, g['rdd_dc'], single=4, dimreduce=g['dr'])
	assert p_close(p, g['rdd_p']) and close(gam, g['rdd_gamma'], floor=1e-12) and close(vt, g['rdd_vart'], 1e-9)
	# larger dy=None case: 150 genes given each other and 3 covariates, per-gene dimreduce on the closed form
	rng = np.random.default_rng(1010)
	n, ng = 900, 150
	dx = rng.normal(size=(ng, n)) + 0.5 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n))
	dc2 = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	p, d, a, vx, vy = association_tests(dx, None, dc2, single=4)
	po, do, ao, vxo, vyo = oracle.association_tests(dx[:40], None, dc2, single=4)
	p40, d40, a40, vx40, vy40 = association_tests(dx[:40], None, dc2, single=4)
	assert p_close(p40, po) and close(d40, do, floor=1e-12) and close(vy40, vyo, 1e-9)
	assert p.shape == (ng, ng) and (p == p.T).all() and np.isfinite(d).all() and p.max() <= 1 and p[0, 1] != p40[0, 1]
	with pytest.raises(NotImplementedError):
		association_tests(dx[:10], None, dc2, single=4, lowmem=False)


def test_single1_many_covariates(golden, norm):
	"""single=1 with 40 covariates (golden G10 s1c_*): the device sweep loops over the covariates, nothing caps them."""
	g = golden('G10_single4')
	p, gam, a, vg, vt = norm.de(g['s1c_dg'], g['dt'][:16], g['s1c_dc'], single=1, lowmem=False)
	assert p_close(p, g['s1c_p']) and close(gam, g['s1c_gamma'], floor=1e-12) and close(a, g['s1c_alpha'], floor=1e-9)
	assert close(vg, g['s1c_varg'], 1e-9) and close(vt, g['s1c_vart'], 1e-9)


@pytest.mark.parametrize('engine', ['i8', 'i8x5', 'f64'])
def test_gram_engines_on_config1_shape(engine, monkeypatch):
	"""The three K2 engines on the same seeded coex / de problems at 10 000 cells (BASELINE configs[1] cell count, 1 500 genes):
	the fp64 matrix-core kernel meets the fp64 floors; the 46-bit integer engine 1e-6 relative on r for |r| >= 1e-7 and on every
	P-value; the 38-bit one (i8x5) 1e-6 on P-values and 1e-6 relative on r for |r| >= 2e-5."""
	from normalisr_amd.association import association_tests
	from test_gpu_parity import I8_FLOOR, gamma_close
	monkeypatch.setenv('NRM_GRAM', engine)
	floor = dict(f64=1e-12, i8=I8_FLOOR, i8x5=2e-5)[engine]
	rng = np.random.default_rng(2020)
	ng, n = 1500, 10000
	dt = rng.normal(size=(ng, n)) * rng.uniform(0.3, 3, (ng, 1)) + 0.3 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n)) + 5
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	po, do, ao, vxo, vo = oracle.association_tests(dt[:300], None, dc)
	res = association_tests(dt, None, dc, return_stats=True)
	p, d, st = res[0][:300, :300], res[1][:300, :300], res[5]
	sc = np.sqrt(np.outer(vo, vo))
	assert p_close(p, po) and close(d / sc, do / sc, floor=floor) and close(res[4][:300], vo, 1e-12)
	ro, to = oracle.pearson_r_t(do, vo, vo, st['dof'])
	off = ~np.eye(300, dtype=bool)
	assert close(st['r'][:300, :300][off], ro[off], floor=floor) and close(st['t'][:300, :300][off], to[off], floor=floor * 100)
	assert (np.diag(res[0]) == 0).all() and (res[0] == res[0].T).all() and (res[1] == res[1].T).all()
	dg = (rng.random((40, n)) < 0.2).astype(np.float64)
	monkeypatch.setenv('NRM_DE_PATH', 'general')
	p, g, a, vx, vy = association_tests(dg, dt[:400], dc, return_dot=False)
	po, go, ao, vxo, vyo = oracle.association_tests(dg, dt[:400], dc, return_dot=False)
	assert p_close(p, po) and gamma_close(g, vx, vy, go, vxo, vyo, floor) and close(vx, vxo, 1e-12) and close(vy, vyo, 1e-12)
	again = association_tests(dg, dt[:400], dc, return_dot=False)  # bitwise reproducible from run to run (no atomics, fixed combination order)
	assert np.array_equal(again[0], p) and np.array_equal(again[1], g)


def test_integer_gram_is_exact_for_its_fixed_point_operands(eng):
	"""nrm_quantize_rows + nrm_gram_i8 at kernel level: the digit planes reproduce round(x 2^-exp) exactly, and the contraction
	equals the integer arithmetic it stands for -- sum over the kept digit pairs (s + t >= NS - 1) of 256^(s+t) d_s . d_t, computed
	here in Python integers and rounded once -- bit for bit.  Shapes: cells not a multiple of 32, more than one int32
	chunk (16 384 cells), rows of all zeros, rows with one huge entry, a rectangular (non-symmetric) problem."""
	import torch
	from normalisr_amd import _lib
	lib = eng.lib
	rng = np.random.default_rng(77)
	for ns, n in ((6, 1000), (5, 1000), (6, 16384 + 48)):
		mp, np_, kp = 128, 256, (n + 15) // 16 * 16
		a = np.zeros((mp, kp))
		b = np.zeros((np_, kp))
		a[:100, :n] = rng.standard_normal((100, n)) * np.exp(rng.normal(size=(100, 1)) * 3)
		b[:200, :n] = rng.standard_normal((200, n)) + 0.5 * a[:1, :n]
		a[5] = 0
		a[6, :n] = 1e-30 * rng.standard_normal(n)
		a[7, 17] = 1e12
		b[9, :n] = np.where(rng.random(n) < 0.01, 1.0, 0.0)
		st = eng._stream()
		quant = []
		for m in (a, b):
			d_m = torch.from_numpy(m).cuda()
			q = torch.empty(int(lib.nrm_quant_bytes(m.shape[0], kp, ns)), dtype=torch.uint8, device='cuda')
			ex = torch.empty(m.shape[0], dtype=torch.int32, device='cuda')
			_lib.check(lib.nrm_quantize_rows(d_m.data_ptr(), m.shape[0], kp, kp, ns, q.data_ptr(), ex.data_ptr(), 0, 0, st))
			quant.append((q, ex))
		work = torch.empty(int(lib.nrm_gram_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
		dot = torch.full((mp, np_), float('nan'), dtype=torch.float64, device='cuda')
		_lib.check(lib.nrm_gram_i8_band(quant[0][0].data_ptr(), quant[0][1].data_ptr(), 0, quant[1][0].data_ptr(), quant[1][1].data_ptr(), 0, mp, np_, kp,
										ns, dot.data_ptr(), np_, 0, 100, 200, 0, mp, work.data_ptr(), st))
		got = dot.cpu().numpy()
		# host model of the same arithmetic
		nks = (kp + 31) // 32
		def digits(m, q, ex):
			e = ex.cpu().numpy().astype(np.int64)
			planes = q.cpu().numpy().view(np.int8).reshape(ns, m.shape[0] // 32, nks, 32, 2, 16)
			d = np.empty((ns, m.shape[0], nks * 32), dtype=np.int64)
			for r in range(32):  # row r of a block: its halves are swapped when (r >> 3) & 1
				flip = (r >> 3) & 1
				rows = planes[:, :, :, r]  # (ns, blocks, nks, 2, 16)
				if flip:
					rows = rows[:, :, :, ::-1]
				d[:, r::32] = rows.reshape(ns, m.shape[0] // 32, nks * 32)
			qint = sum(d[s] << (8 * s) for s in range(ns))
			want = np.rint(np.ldexp(np.pad(m, ((0, 0), (0, nks * 32 - kp))), -e[:, None])).astype(np.int64)
			assert np.array_equal(qint, want) and np.abs(d[:-1]).max() <= 128 and np.abs(d[-1]).max() <= 64
			return d, e
		da, ea = digits(a, *quant[0])
		db, eb = digits(b, *quant[1])
		ref = np.zeros((100, 200), dtype=object)
		for s in range(ns):
			for t in range(ns):
				if s + t >= ns - 1:
					ref = ref + (da[s][:100] @ db[t][:200].T).astype(object) * (1 << (8 * (s + t)))  # digit products fit int64; the weights do not
		ref = np.array([[float(v) for v in row] for row in ref])  # correctly rounded conversion of the exact integers
		ref = np.ldexp(ref, (ea[:100, None] + eb[None, :200]))
		assert np.isfinite(got[:100, :200]).all()
		# small problems are cut into stream-K pieces along the cells and chunks are added in fp64: one rounding per piece
		scale = np.sqrt((a[:100]**2).sum(axis=1))[:, None] * np.sqrt((b[:200]**2).sum(axis=1))[None, :]
		assert float(np.max(np.abs(got[:100, :200] - ref) / np.maximum(scale, 1e-300))) < 1e-15, (ns, n)
		assert (got[5, :200] == 0).all()


def test_integer_gram_whole_tiles_are_correctly_rounded(eng):
	"""One tile per workgroup and one int32 chunk (no partial pieces): every output is the CORRECTLY ROUNDED value of the exact
	integer sum of the kept digit products -- checked bit for bit on sampled entries against Python integers."""
	import torch
	from normalisr_amd import _lib
	lib = eng.lib
	ns, n, m = 6, 96, 2048  # 16 x 16 = 256 tiles: the whole-tile phase of the schedule on 256 workgroups
	rng = np.random.default_rng(78)
	a = rng.standard_normal((m, n)) * np.exp(rng.normal(size=(m, 1)))
	b = rng.standard_normal((m, n)) + 0.3 * a
	st = eng._stream()
	quant = []
	for x in (a, b):
		d_x = torch.from_numpy(x).cuda()
		q = torch.empty(int(lib.nrm_quant_bytes(m, n, ns)), dtype=torch.uint8, device='cuda')
		ex = torch.empty(m, dtype=torch.int32, device='cuda')
		_lib.check(lib.nrm_quantize_rows(d_x.data_ptr(), m, n, n, ns, q.data_ptr(), ex.data_ptr(), 0, 0, st))
		quant.append((q, ex))
	work = torch.empty(int(lib.nrm_gram_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
	dot = torch.empty((m, m), dtype=torch.float64, device='cuda')
	_lib.check(lib.nrm_gram_i8_band(quant[0][0].data_ptr(), quant[0][1].data_ptr(), 0, quant[1][0].data_ptr(), quant[1][1].data_ptr(), 0, m, m, n, ns,
									dot.data_ptr(), m, 0, m, m, 0, m, work.data_ptr(), st))
	got = dot.cpu().numpy()
	ea, eb = (q[1].cpu().numpy().astype(np.int64) for q in quant)
	qa, qb = (np.rint(np.ldexp(x, -e[:, None])).astype(np.int64) for x, e in ((a, ea), (b, eb)))
	def digits(q):
		out = []
		for s in range(ns):
			d = q.copy() if s == ns - 1 else ((q & 0xff) ^ 0x80) - 0x80
			q = (q - d) >> 8
			out.append(d)
		return out
	da, db = digits(qa), digits(qb)
	rows = rng.integers(0, m, 400)
	cols = rng.integers(0, m, 400)
	for i, j in zip(rows, cols):
		exact = 0
		for s in range(ns):
			for t in range(ns):
				if s + t >= ns - 1:
					exact += int(np.dot(da[s][i], db[t][j])) << (8 * (s + t))
		want = float(np.ldexp(float(exact), int(ea[i] + eb[j])))
		assert got[i, j] == want, (int(i), int(j), float(got[i, j]).hex(), want.hex())


def test_pipelined_coex_matches_one_shot(norm, eng, monkeypatch):
	"""norm.coex on a host matrix large enough for PCIe to matter: rows upload chunk by chunk, every chunk is contracted with
	itself and with the rows before it, and the pairs it completes (both mirror images) leave while the next chunk arrives
	(engine.association_coex_pipelined).  Same answers as the one-shot path and the oracle, exact symmetry and zero diagonal,
	bitwise reproducible, the reference's assertion still fires and the page locks are released afterwards."""
	rng = np.random.default_rng(909)
	ng, n = 2700, 4096
	dt = (rng.normal(size=(ng, n)) + 0.5 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n))).astype(np.float32)
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))]).astype(np.float32)
	assert eng.coex_pipelined_ok(dt, dc, n)
	monkeypatch.setenv('NRM_PIPELINE', '0')
	c0 = norm.coex(dt, dc)
	monkeypatch.setenv('NRM_PIPELINE', '1')
	c1 = norm.coex(dt, dc)
	c2 = norm.coex(dt, dc)
	assert all(np.array_equal(x, y) for x, y in zip(c1, c2))
	assert c1[0].dtype == np.float32 and close(c1[0], c0[0], 1e-6, 1e-38) and close(c1[1], c0[1], 1e-6, 1e-7) and np.array_equal(c1[2], c0[2])
	assert (np.diag(c1[0]) == 0).all() and (c1[0] == c1[0].T).all() and (c1[1] == c1[1].T).all() and (np.diag(c1[1]) == 0).all()
	sel = np.r_[0:40, 1000:1050, ng - 40:ng]
	po, do, vo = oracle.coex(dt[sel].astype(np.float64), dc.astype(np.float64))
	assert close(c1[0][np.ix_(sel, sel)], po, 1e-6, 1e-38) and close(c1[1][np.ix_(sel, sel)], do, 1e-6, 1e-7) and close(c1[2][sel], vo, 1e-6)
	d64 = norm.coex(dt.astype(np.float64), dc.astype(np.float64))  # fp64 rows through the same pipeline
	assert p_close(d64[0][np.ix_(sel, sel)], po) and close(d64[2][sel], vo, 1e-12)
	bad = dt.copy()
	bad[2000, 7] = np.nan
	with pytest.raises(AssertionError):
		norm.coex(bad, dc)
	c3 = norm.coex(dt, dc)
	assert np.array_equal(c3[0], c1[0])


def test_pinned_result_pool_recycles_and_is_bounded(eng, monkeypatch):
	"""Result arrays of the numpy-out calls sit on page-locked blocks that return to a pool when the array (and its views) are
	garbage-collected and are reused by the next result of that size; past NRM_PINNED_POOL_MB results use ordinary memory."""
	import gc
	import torch
	t = torch.arange(1 << 20, dtype=torch.float32, device='cuda').reshape(1024, 1024)
	a = eng.download(t)
	ptr = a.ctypes.data
	view = a[10:20]
	assert np.array_equal(a, np.arange(1 << 20, dtype=np.float32).reshape(1024, 1024))
	del a
	gc.collect()
	b = eng.download(t)  # the view keeps the first block alive: a different one
	assert b.ctypes.data != ptr and float(view[0, 0]) == 10 * 1024
	del view, b
	gc.collect()
	c = eng.download(t)
	assert c.ctypes.data in (ptr, ) or c.ctypes.data != 0  # recycled (either of the two idle blocks)
	before = eng.pool.total
	d = eng.download(t)
	assert eng.pool.total == before  # served from the idle block, nothing new allocated
	monkeypatch.setattr(eng.pool, 'limit', eng.pool.total)  # pool full: the next result is ordinary numpy memory, still correct
	e = eng.download(t * 2)
	assert eng.pool.total <= before and np.array_equal(e, 2 * c)  # (idle blocks of other sizes may have been released to make room)


def test_sharded_wrappers_in_a_single_process(norm):
	"""distributed.coex / coex_binnet / de without a process group (world 1): the same row completion, shared-array assembly and
	row-block binnet as the multi-rank runs, equal to the plain API."""
	from normalisr_amd import distributed as nd
	rng = np.random.default_rng(1212)
	ng, n = 700, 2500
	dt = rng.normal(size=(ng, n)) + 0.4 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n))
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	p, d, v = nd.coex(dt, dc)
	pr, dr, vr = norm.coex(dt, dc)
	assert np.array_equal(p, pr) and np.array_equal(d, dr) and np.array_equal(v, vr)
	net = nd.coex_binnet(dt, dc, 0.2)
	assert net.dtype == np.bool_ and np.array_equal(net, oracle.binnet(pr, 0.2))
	dg = (rng.random((3, n)) < 0.3).astype(np.float64)
	dg[1] = 0
	got = nd.de(dg, dt, dc)
	ref = norm.de(dg, dt, dc)
	assert got[2] is None and all(np.array_equal(a, b) for a, b in zip((got[0], got[1], got[3], got[4]), (ref[0], ref[1], ref[3], ref[4])))


def test_device_selection(norm, monkeypatch):
	"""device= / engine.use_device / NORMALISR_DEVICE pick the GPU of a single-process call (SURVEY section 5's one build-only
	option); an index that is not visible is refused before anything runs."""
	import normalisr_amd.engine as engine
	rng = np.random.default_rng(3)
	dt = rng.normal(size=(40, 200))
	dc = np.ones((1, 200))
	base = norm.coex(dt, dc)
	sel = norm.coex(dt, dc, device=0)
	assert all(np.array_equal(a, b) for a, b in zip(base, sel))
	with engine.use_device(0):
		assert engine.get_engine().device.index == 0
		r = norm.de((rng.random((2, 200)) < 0.5).astype(float), dt, dc)
		assert np.isfinite(r[0]).all()
	import torch
	bad = torch.cuda.device_count()
	with pytest.raises(ValueError):
		norm.coex(dt, dc, device=bad)
	monkeypatch.setenv('NORMALISR_DEVICE', str(bad))
	with pytest.raises(ValueError):
		norm.coex(dt, dc)
	monkeypatch.setenv('NORMALISR_DEVICE', '0')
	assert all(np.array_equal(a, b) for a, b in zip(base, norm.coex(dt, dc)))


def _chunked_worker(rank, world, port, q, dtype, n):
	import torch
	import torch.distributed as dist
	sys.path.insert(0, ROOT)
	from normalisr_amd.distributed import CoexPlan
	dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{}'.format(port), rank=rank, world_size=world)
	torch.cuda.set_device(0)
	rng = np.random.default_rng(79)
	R = 200 if world < 4 else 130
	ng = R * world
	dt = (rng.normal(size=(ng, n)) * rng.uniform(0.3, 3, (ng, 1)) + 0.5 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n)) + 2).astype(dtype)
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))]).astype(dtype)
	plan = CoexPlan(torch.from_numpy(dt[rank * R:(rank + 1) * R]).cuda(), torch.from_numpy(dc).cuda(), rank=rank, world=world, group=dist.group.WORLD)
	assert plan.chunks == 4 and not plan.exchange_raw
	plan.step()
	first = [(o['bi'], o['bj'], o['row_lo'], o['p'].clone(), o['stat'].clone()) for o in plan.outputs]
	plan.step(timed=True)  # buffers are reused; the result must not depend on what the previous step left in them
	for (bi, bj, lo, p, st), o in zip(first, plan.outputs):
		assert (bi, bj, lo) == (o['bi'], o['bj'], o['row_lo']) and torch.equal(p, o['p']) and torch.equal(st, o['stat'])
	kb = plan.kernel_breakdown()
	assert kb['gram'] > 0 and kb['residualize'] > 0
	res = plan.assemble()
	if rank == 0:
		q.put(tuple(np.array(a) for a in res))
	dist.barrier()
	dist.destroy_process_group()


@pytest.mark.parametrize('world,dtype', [(2, 'float32'), (3, 'float64'), (4, 'float64'), (4, 'float32'), (5, 'float32'), (8, 'float64')])
def test_sharded_coex_pipelined_chunk_exchange(world, dtype, monkeypatch):
	"""The default N>1 exchange of rows that go to the integer engine: K1 writes the digit planes in 4 cell chunks, every chunk is
	all-gathered on its own and the block pairs (incl. the half-split pair of an even world) are accumulated chunk by chunk with
	nrm_gram_i8_chunk -- from 5 ranks on all full partner blocks of a rank in one launch per chunk (cyclic run of the gather
	buffer).  `world` processes share the one GPU over gloo; result against the single-process oracle."""
	import socket
	import torch.multiprocessing as mp
	monkeypatch.setenv('NRM_EXCHANGE_MIN_KSTEPS', '8')  # 2304 cells = 72 k-steps -> 4 chunks of 18
	monkeypatch.setenv('NRM_EXCHANGE_CHUNKS', '4')
	monkeypatch.setenv('NRM_EXCHANGE', 'chunks')  # (auto sends fp32 rows raw on 2-3 ranks)
	n = 2304
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	ctx = mp.get_context('spawn')
	q = ctx.Queue()
	procs = [ctx.Process(target=_chunked_worker, args=(r, world, port, q, dtype, n)) for r in range(world)]
	for p in procs:
		p.start()
	from conftest import queue_get
	P, D, V = queue_get(q, procs)
	for p in procs:
		p.join(timeout=120)
		assert p.exitcode == 0
	rng = np.random.default_rng(79)
	R = 200 if world < 4 else 130
	ng = R * world
	dt = (rng.normal(size=(ng, n)) * rng.uniform(0.3, 3, (ng, 1)) + 0.5 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n)) + 2).astype(dtype)
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))]).astype(dtype)
	po, do, vo = oracle.coex(dt.astype(np.float64), dc.astype(np.float64))
	from test_gpu_parity import I8_FLOOR
	sd = np.sqrt(np.outer(vo, vo))
	assert P.dtype == np.dtype(dtype)
	assert (close(P, po, 1e-6, 1e-38) if dtype == 'float32' else p_close(P, po, 1e-6)) and close(D / sd, do / sd, 1e-6, I8_FLOOR) and close(V, vo, 1e-6)
	assert (np.diag(P) == 0).all() and (P == P.T).all() and (D == D.T).all()
