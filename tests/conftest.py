import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
	config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
	def load(name):
		return np.load(os.path.join(GOLDEN, name + '.npz'))
	return load


def relerr(a, b, atol=0.):
	"""max |a-b| / (|b| + atol-scaled floor), elementwise relative error with absolute floor."""
	a = np.asarray(a, dtype=np.float64)
	b = np.asarray(b, dtype=np.float64)
	return float(np.max(np.abs(a - b) / (np.abs(b) + atol))) if a.size else 0.


# ---- collection order of the GPU suite ---------------------------------------------------------------------------------------
# The driver runs `pytest tests -x -q -m gpu`: one failure ends the run.  What proves parity runs first -- the reference-held
# fixtures (every test that takes the `golden` fixture, G1-G15) and the kernels' exactness tests --, then the C-ABI entries, then
# the rest of the parity tests, then everything that starts other processes (bench.py, the sharded command line), and LAST the
# timing bounds (tests/test_zz_perf_gpu.py).  Inside a tier the order of the files is kept.
_FIRST = ('golden', 'g11_', 'g13_', 'g14_', 'integer_gram', 'gram_engines', 'gram_kernel_layout', 'pvalue_kernel_table', 'pvalue_function_against_mpmath',
		  'block_g7', 'k1_row_records', 'single5_with_a_mask', 'full_size_c2_properties')
_ENTRIES = ('host_entry', 'c_entry', 'c_entries', 'without_torch', 'without_importing_torch')
_PROCESSES = ('bench_', 'sharded', 'two_ranks', 'rccl_', 'cli_')


def _tier(item):
	name = item.name.lower()
	if 'test_zz_perf' in item.nodeid:
		return 9
	if 'golden' in getattr(item, 'fixturenames', ()) or any(k in name for k in _FIRST):
		return 0
	if any(k in name for k in _ENTRIES):
		return 1
	if any(k in name for k in _PROCESSES) or 'bench_default' in getattr(item, 'fixturenames', ()):
		return 3
	return 2


def pytest_collection_modifyitems(config, items):
	gpu = [i for i, it in enumerate(items) if it.get_closest_marker('gpu') is not None]
	ordered = sorted((items[i] for i in gpu), key=_tier)  # stable: file order inside a tier
	for i, it in zip(gpu, ordered):
		items[i] = it


def run_bench(extra_args, env_extra, timeout=900):
	"""`python bench.py ...` as the driver runs it (a child process); returns the contract line (the LAST line of stdout, under 4 KB) with the
	full record of every workload, printed on the lines before it, under '_detail'."""
	import json
	import subprocess
	env = dict(os.environ)
	env.update(env_extra)
	r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra_args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
					   timeout=timeout)
	assert r.returncode == 0, r.stderr[-3000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
	assert lines and r.stdout.rstrip().splitlines()[-1] == lines[-1], r.stdout[-2000:]  # the contract line is the LAST line
	assert len(lines[-1]) < 4096, len(lines[-1])  # (the driver keeps the tail of stdout: round 4's 14 KB line lost its first extras there)
	out = json.loads(lines[-1])
	detail = {}
	for ln in lines[:-1]:  # the full record of every workload, printed before it
		d = json.loads(ln)
		detail[d.pop('workload_detail')] = d
	out['_detail'] = detail
	return out


@pytest.fixture(scope='session')
def bench_default():
	"""One default N=1 run of bench.py (short: 3 steps, no CPU baseline, no PCIe leg), shared by the test of the line's CONTENT
	(tests/test_gpu_round2.py) and the timing bounds (tests/test_zz_perf_gpu.py)."""
	return run_bench(['--steps', '3', '--warmup', '1', '--cpu-seconds', '0', '--e2e', '0', '--extras-steps', '2'], {})


def queue_get(q, procs, timeout=300):
	"""q.get() that does not sit out its timeout when a child has already died: a rank that fails at start (a bad import, no HBM left for it) would
	otherwise cost the suite five minutes per test -- round 6's first run on the GPU box spent 35 of its 40 minutes in seven such waits (a conftest.py
	added at the repo root shadowed this file in the spawned ranks, which import `conftest` by name)."""
	import queue
	import time
	t0 = time.monotonic()
	while True:
		try:
			return q.get(timeout=2)
		except queue.Empty:
			dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
			if dead:
				for p in procs:
					if p.is_alive():
						p.terminate()
				raise AssertionError('rank process(es) died before answering: (rank, exit code) {}'.format(dead))
			if time.monotonic() - t0 > timeout:
				for p in procs:
					if p.is_alive():
						p.terminate()
				raise AssertionError('no answer from the rank processes in {} s'.format(timeout))


@pytest.fixture(autouse=True)
def _free_device_caches_before_process_tests(request):
	"""Tests that start OTHER processes on this box's one GPU (sharded runs over gloo, bench.py, the command line) are collected last, after the large
	single-process tests: what this process cached meanwhile (torch's caching allocator after the configs[4] tests holds a large part of the 288 GB; the
	library's scratch pool and upload ring) is handed back first, so that the ranks' own allocations do not depend on what ran before them."""
	if request.node.get_closest_marker('gpu') is not None and _tier(request.node) == 3:
		try:
			import torch
			if torch.cuda.is_available():
				torch.cuda.synchronize()
				torch.cuda.empty_cache()
			from normalisr_amd import _lib
			lib = _lib.load()
			lib.nrm_release_cache()
			lib.nrm_upload_release()
		except Exception:  # noqa: BLE001 -- best effort: the test itself reports what is wrong
			pass
	yield
