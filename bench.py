#!/usr/bin/env python3
"""bench.py -- association tests/sec on the BASELINE.json workloads.

Headline (`value`): BASELINE.json configs[1] -- norm.coex gene x gene on 5k genes x 10k cells, fp32 input,
3 covariates (2 random + intercept), seeded synthetic data.  A step is one full pass of
the hot path over the matrix resident in HBM: residualise + quantise (K1) -> Gram contraction on the int8 matrix cores, exact
for its 46-bit fixed-point operands (K2, k_gram_i8; problems under 2048 cells: the fp64-MFMA kernel k_gram_f64) -> per-pair
sweep R^2 -> p, covariance with the integer engine's correction and accuracy guard (K3); outputs stay in HBM.  tests = ng(ng-1)/2 unique pairs.

For N>1 (one process per GPU, RCCL) the headline is the workload BASELINE.json's 8-GPU target is quoted on: configs[4], norm.coex
on 3750 gene rows per rank x 500 000 cells fp64 (N=8: the full 30 000 x 30 000 problem) -- gene-row blocks residualised locally,
their digit planes exchanged by all-gather in cell chunks; the rank count seen by the collective, the bytes a rank receives and the
exchange time compute did not hide are in the line.  configs[1] weak-scaled (genes ~ sqrt(N)) moves to `extra_workloads.coex_c2` there.

The same JSON line carries, under `extra_workloads`, the other BASELINE configs measured the same way (each
with its own ms_per_step and roofline): de_c3 = configs[2] (1 x 20k genes x 100k cells, 20 covariates:
HBM-bound streaming kernel), de_c4 = configs[3] (1k gRNAs x 15k genes x 50k cells, gene rows sharded over the
ranks, no collective), coex_c5 = the per-rank shape of configs[4] (3750 gene rows per rank x 500k cells,
fp64; at N=8 that is the full 30k x 30k problem, residual blocks exchanged by all-gather), de_c4_single4 / de_c4_single1 = the
CRISPR screen of configs[3] as the reference's example runs it besides the naive test (`de -m covariate` / `-m single`,
examples/GSE120861/code/cmd_highmoi.sh:19-22), coex_c2_f64 / coex_c5_f64 = configs[1] / the configs[4] per-rank slice on the fp64 matrix cores
(NRM_GRAM=f64: the dtype the north star names literally), binnet_c5 = binnet on a 30 000 x 30 000 fp64 P-value matrix (the consumer of configs[4]'s output), normvar_c2 = norm.normvar on a
configs[1]-sized matrix resident in HBM (the step in front of the hot path), chain_c2 = normvar -> coex -> binnet as one resident chain.
`--workload X` makes X the headline instead; `--no-extras` skips them.

Launch: `python bench.py --gpus N` starts its own N ranks (torch.distributed.run on 127.0.0.1) when it is
not already running under one; under `python -m torch.distributed.run ... bench.py --gpus N` it uses the
ranks it was given.  Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_gram_i8),
timed live with HIP events on the launch stream; `cpu_baseline` times the CPU oracle (a port of the
reference's algorithm, test infrastructure) on a bounded sample on this box's host cores (N=1 only); the extra
workloads carry their own, extrapolated from sampled rows.  `kernels_roofline` prices the two kernels beside K2 against the rooflines that
bound them (K1: HBM; K3: fp64 vector ALU).  `guard` is the verdict of the integer engine's accuracy
guard over the timed steps (pairs it could not certify: must be 0 here).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same commands (tools/profile_r06.sh), newest first
PMC_FILES = {w: [os.path.join(ROOT, 'profiles', f) for f in fs] for w, fs in dict(
	coex_c2=('r06_pmc_c2.json', 'r05_pmc_c2.json', 'r04_pmc_c2.json', 'r03_pmc_c2.json', 'r02_pmc_c2.json', 'r01_pmc_c2.json'), de_c3=('r06_pmc_de_c3.json', 'r05_pmc_de_c3.json', 'r04_pmc_de_c3.json', 'r03_pmc_de_c3.json', 'r02_pmc_de_c3.json'),
	de_c4=('r06_pmc_de_c4.json', 'r05_pmc_de_c4.json', 'r04_pmc_de_c4_sparse.json'), de_c4_dense=('r04_pmc_de_c4.json', 'r03_pmc_de_c4.json'), coex_c5=('r06_pmc_coex_c5.json', 'r05_pmc_coex_c5.json', 'r04_pmc_coex_c5.json', 'r03_pmc_coex_c5.json'),
	de_c4_single4=('r06_pmc_de_c4_single4.json', 'r05_pmc_de_c4_single4.json', 'r04_pmc_de_c4_single4_sparse.json'), de_c4_single4_dense=('r04_pmc_de_c4_single4.json', ),
	de_c4_single1=('r06_pmc_de_c4_single1.json', 'r05_pmc_de_c4_single1.json', 'r04_pmc_de_c4_single1.json'), binnet_c5=('r06_pmc_binnet_c5.json', 'r05_pmc_binnet_c5.json', 'r04_pmc_binnet_c5.json'),
	normvar_c2=('r06_pmc_normvar_c2.json', 'r05_pmc_normvar_c2.json')).items()}


def pmc_traffic(workload, roof, kernels=None):
	"""HBM-side bytes per launch of the roofline's kernel from the committed PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE).
	kernels: the kernels whose launches make up roof['kernel_ms'] (default: the one roof['kernel'] names) -- their bytes are added; a
	summary that lacks one of them is not used (the traffic of ANOTHER kernel is never attached: round-4 verdict)."""
	wants = list(kernels) if kernels else [roof['kernel'].split(' ')[0]]
	for f in PMC_FILES.get(workload, ()):
		try:
			with open(f) as fh:
				pm = json.load(fh)
			total, used, clock = 0.0, [], None
			for want in wants:
				cands = [(k, v) for k, v in pm.items() if k.split('<')[0].split(' ')[0] == want and 'hbm_bytes_per_launch' in v]
				if not cands:
					break
				k, entry = max(cands, key=lambda kv: kv[1]['hbm_bytes_per_launch'])  # (the long dispatch class of a kernel launched on two sizes)
				total += entry['hbm_bytes_per_launch']
				used.append(k)
				clock = entry.get('effective_clock_ghz', clock) if want == wants[0] else clock
			else:
				roof['traffic'] = total
				if clock is not None:  # the chip clocks down under int8 MFMA load: the nominal peak assumes 2.4 GHz
					roof['effective_clock_ghz_profiled'] = round(clock, 3)
				roof['traffic_unit'] = 'bytes/launch'
				roof['traffic_source'] = '{} [{}] (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE, separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)'.format(
					os.path.relpath(f, ROOT), ' + '.join(used))
				return
		except (OSError, KeyError, ValueError):
			continue
	roof['traffic'] = None
	roof.pop('traffic_source', None)


F64_MFMA_PEAK_TFLOPS = 78.6  # v_mfma_f64_16x16x4_f64: 32 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz (= 1/2 of the 157.3 TF fp32 matrix peak of MI355X_MICROARCH.md)
F32_MFMA_PEAK_TFLOPS = 157.3  # the roofline BASELINE.json's north_star names
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA: 2x the bf16 rate per clock (MI355X_MICROARCH.md, matrix cores): 2 x 2.5 PF


def gram_roofline(n_cells, flops, gram_ms, rows_pad, k_pad):
	"""Roofline entry of the dominant kernel (K2).  `achieved` is ALGORITHMIC work (2 n_cell flop per test) over the
	kernel's measured duration.  The integer engine (>= 2048 cells) spends nslices (nslices + 1) / 2 int8 multiply-adds per
	fp64-equivalent one, so its peak is the dense int8 MFMA peak divided by that count; the fp64 kernel is priced against the
	fp64 MFMA peak.  Both are also shown against the fp32 MFMA peak the north star names."""
	ns = SLICES(n_cells)
	achieved = flops / (gram_ms * 1e-3) / 1e12
	if ns:
		pairs = ns * (ns + 1) // 2
		peak = I8_MFMA_PEAK_TOPS / pairs
		kern, note = 'k_gram_i8', ('exact fixed-point contraction: {}-bit operands in {} int8 digit planes, {} digit products per multiply-add on '
								   'v_mfma_i32_32x32x32_i8, int32 accumulation, fp64 combine; peak = {:.0f} TOP/s dense int8 MFMA / {}').format(
									   8 * ns - 2, ns, pairs, I8_MFMA_PEAK_TOPS, pairs)
		abytes = float(2 * ns) * rows_pad * k_pad  # both operand panels, one byte per digit
	else:
		peak, kern, note, abytes = F64_MFMA_PEAK_TFLOPS, 'k_gram_f64', 'fp64 matrix cores (v_mfma_f64_16x16x4_f64)', 16.0 * rows_pad * k_pad
	return dict(bound='mfma', kernel=kern, achieved=achieved, peak=peak, unit='TFLOP/s', frac=achieved / peak, traffic=None,
				algorithmic_bytes=abytes, kernel_ms=gram_ms, arithmetic=note, int8_tops_executed=(achieved * (ns * (ns + 1) // 2) if ns else None),
				frac_of_fp64_mfma_peak=achieved / F64_MFMA_PEAK_TFLOPS, frac_of_fp32_mfma_peak=achieved / F32_MFMA_PEAK_TFLOPS)
HBM_PEAK_GBS = 8000.0
F64_VALU_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz


def side_rooflines(kernels_ms, rows, n, esz, ns, tests, out_esz):
	"""The two kernels beside K2, each against the roofline that bounds it, from ALGORITHMIC work over the measured
	time.  K1 is HBM-bound: it reads every row once (its second sweep is served from L2 / MALL for rows that fit there) and writes
	ns digit bytes per value (or the fp64 residual).  K3 is bound by the fp64 vector ALU (about 250 fp64 instructions per test out of
	~500 in all: P-value function, correction, guard); its traffic -- 8 bytes read, 2 results written per test -- is reported beside it."""
	out = {}
	k1, k3 = kernels_ms.get('residualize'), kernels_ms.get('sweep')
	if k1:
		b = float(rows) * n * (esz + (ns if ns else 8))
		out['k_residualize'] = dict(bound='hbm', algorithmic_bytes=b, achieved=b / (k1 * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
									frac=b / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS, kernel_ms=k1)
	if k3:
		b = float(tests) * (8 + 2 * out_esz)
		out['k_assoc_sweep'] = dict(bound='fp64 valu', tests_per_s=tests / (k3 * 1e-3), fp64_instructions_per_test_estimate=250,
									achieved=250 * tests / (k3 * 1e-3) / 1e12, peak=F64_VALU_PEAK_TFLOPS / 2, unit='T fp64 instructions/s',
									frac=250 * tests / (k3 * 1e-3) / 1e12 / (F64_VALU_PEAK_TFLOPS / 2), algorithmic_bytes=b,
									hbm_gbs=b / (k3 * 1e-3) / 1e9, kernel_ms=k3)
	return out


C5_ROWS_PER_RANK = 3750  # configs[4]: 30 000 genes over 8 GPUs
C5_CELLS = 500000


def synth_c2(ng, n, seed, device, torch, row0=0, dtype=None, loading=0.3):
	"""N(0,1) + loading * gene loading * shared latent factor; dc = [2 x N(0,1); ones]."""
	dtype = torch.float32 if dtype is None else dtype
	g = torch.Generator(device=device)
	g.manual_seed(seed)
	lat = torch.randn((1, n), generator=g, device=device, dtype=dtype)
	dc = torch.cat([torch.randn((2, n), generator=g, device=device, dtype=dtype),
					torch.ones((1, n), device=device, dtype=dtype)])
	g2 = torch.Generator(device=device)
	g2.manual_seed(seed * 1000003 + row0)
	load = torch.randn((ng, 1), generator=g2, device=device, dtype=dtype)
	dt = torch.randn((ng, n), generator=g2, device=device, dtype=dtype)
	dt.addcmul_(load, lat, value=loading)  # in place: no second matrix-sized temporary (C5 rows are 15 GB per rank)
	return dt, dc


def cpu_baseline_worker(ng, n_cells, nc, seed, min_seconds, extras_too=False):
	"""Runs in a child process started with BLAS threads pinned to 1 (the reference launcher's convention,
	bin/normalisr:3) and times the CPU oracle's tile loop with nth = all cores on the full workload."""
	import oracle
	rng = np.random.default_rng(seed)
	lat = rng.normal(size=(1, n_cells))
	dc = np.vstack([rng.normal(size=(nc - 1, n_cells)), np.ones((1, n_cells))])
	cores = os.cpu_count() or 1
	dt = rng.normal(size=(ng, n_cells)) + 0.3 * rng.normal(size=(ng, 1)) * lat
	oracle.coex(dt[:64], dc)  # warm-up (library load)
	reps, t0 = 0, time.perf_counter()
	while True:
		if extras_too and min_seconds <= 0:  # (the extras' child: the headline was timed by the first one)
			reps, el = 1, 1.0
			break
		oracle.coex(dt, dc, nth=cores)
		reps += 1
		el = time.perf_counter() - t0
		if el >= min_seconds or reps >= 50:
			break
	pairs = ng * (ng - 1) // 2
	out = dict(value=pairs * reps / el, unit='tests/s', cores=cores, kind='port',
			   sample='{} pass(es) of coex on {} genes x {} cells fp64 ({} pairs each) in {:.1f} s; CPU oracle tile loop '
			   '(500x500 tiles, per-tile residualisation as association.py:224-249), thread pool nth={}, BLAS threads=1'.format(
				   reps, ng, n_cells, pairs, el, cores))
	del dt
	extras = {}
	if extras_too:
		# the other BASELINE configs on SAMPLED rows (the full problems take minutes to hours on the host): rate per test of the
		# sample, flagged as extrapolated -- the per-test cost of the reference's algorithm does not depend on how many tiles follow
		def timed(fn, tests, what):
			t0 = time.perf_counter()
			fn()
			el = time.perf_counter() - t0
			return dict(value=tests / el, unit='tests/s', cores=cores, kind='port', extrapolated=True, sample='{} in {:.1f} s; thread pool nth={}, BLAS threads=1'.format(what, el, cores))
		r = np.random.default_rng(3)
		n3, g3 = 100000, min(20000, 500 * max(1, min(cores, 4)))
		dc3 = np.vstack([r.normal(size=(19, n3)), np.ones((1, n3))])
		dg3 = (r.random((1, n3)) < 0.5).astype(np.float64)
		dt3 = r.normal(size=(g3, n3))
		extras['de_c3'] = timed(lambda: oracle.de(dg3, dt3, dc3, nth=cores), g3, 'norm.de 1 x {} of the 20000 genes x {} cells, 20 covariates'.format(g3, n3))
		del dt3, dc3
		n4, g4 = 50000, 500 * max(1, min(cores // 2, 2))
		dc4 = np.vstack([r.normal(size=(4, n4)), np.ones((1, n4))])
		dg4 = (r.random((1000, n4)) < 0.01).astype(np.float64)
		dg1 = (r.random((1000, n4)) < 0.001).astype(np.float64)  # (single=1: low MOI, see bench_de_method)
		dt4 = r.normal(size=(g4, n4))
		extras['de_c4'] = timed(lambda: oracle.de(dg4, dt4, dc4, nth=cores), 1000 * g4, 'norm.de 1000 gRNAs x {} of the 15000 genes x {} cells, 5 covariates'.format(g4, n4))
		# the CRISPR screen as the reference's example runs it (cmd_highmoi.sh:19-22), on 100 of the 1000 gRNAs: both methods cost the same
		# per tested gRNA whatever their number as long as the others are there -- single=4: the other 900 rows join the covariates (the
		# same model for the 100 tested, and the same (groupings + covariates - 1)^2 SVD per grouping); single=1: the cells that carry one
		# of the other 900 are dropped beforehand (what association.py:915-916 does with them for every tested row).  single=4's SVDs do
		# not depend on the gene count, so it is timed at two gene counts: t = t_groupings + genes * t_gene.
		nx4 = 100
		others = dg4[nx4:]
		dc4s = np.vstack([others, dc4])
		keep1 = dg1[nx4:].sum(axis=0) == 0
		dg1s, dt1s, dc1s = np.ascontiguousarray(dg1[:nx4, keep1]), None, np.ascontiguousarray(dc4[:, keep1])

		def de4(g, single):
			t0 = time.perf_counter()
			if single == 4:
				oracle.de(dg4[:nx4], dt4[:g], dc4s, single=4, nth=cores)
			else:
				oracle.de(dg1s, np.ascontiguousarray(dt4[:g][:, keep1]), dc1s, single=1, nth=cores)
			return time.perf_counter() - t0
		for name, single in (('de_c4_single4', 4), ('de_c4_single1', 1)):
			try:
				ga, gb = (g4 // 4, g4 // 2) if single == 4 else (g4 // 8, g4 // 4)
				ta, tb = de4(ga, single), de4(gb, single)
				t_gene = max(tb - ta, 1e-9) / (gb - ga)
				t_fix = max(ta - t_gene * ga, 0.0)
				full = (t_fix + 15000 * t_gene) * (1000 / nx4)
				extras[name] = dict(value=1000 * 15000 / full, unit='tests/s', cores=cores, kind='port', extrapolated=True,
									sample='norm.de(single={}) on {} of the 1000 gRNAs (the others kept in the model) x {} and {} of the 15000 genes x {} cells in {:.1f} + {:.1f} s; per-grouping part {:.2f} s, per gene {:.2e} s; thread pool nth={}, BLAS threads=1'.format(
										single, nx4, ga, gb, n4, ta, tb, t_fix, t_gene, cores))
			except Exception as e:  # noqa: BLE001
				extras[name] = dict(error='{}: {}'.format(type(e).__name__, e))
		del others, dc4s, dg1s, dc1s
		del dt4, dg4, dc4
		n5, g5 = 500000, 264  # (the reference's tile size at this shape is 132)
		dc5 = np.vstack([r.normal(size=(2, n5)), np.ones((1, n5))])
		dt5 = r.normal(size=(g5, n5))
		extras['coex_c5'] = timed(lambda: oracle.coex(dt5, dc5, nth=cores), g5 * (g5 - 1) // 2, 'norm.coex on {} of the 30000 genes x {} cells fp64'.format(g5, n5))
	out['extras'] = extras
	print(json.dumps(out))


def cpu_baseline(ng, n_cells, nc, seed, min_seconds=10.0, extras=False):
	"""The headline's CPU baseline and -- in a second child, bounded at 180 s, whose failure costs only the extras' baselines -- those of
	the extra workloads."""
	import subprocess
	env = dict(os.environ)
	for k in ('OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS', 'OMP_NUM_THREADS'):
		env[k] = '1'
	env['HIP_VISIBLE_DEVICES'] = ''

	def child(secs, with_extras, limit):
		r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(ng), str(n_cells), str(nc), str(seed), str(secs), str(int(with_extras))],
						   env=env, stdout=subprocess.PIPE, text=True, timeout=limit)
		return json.loads(r.stdout.strip().splitlines()[-1])
	out = child(min_seconds, False, 600)
	if extras:
		try:
			out['extras'] = child(0.0, True, 180).get('extras', {})
		except Exception as e:  # noqa: BLE001 -- reported, never fatal
			out['extras'] = dict(_error='{}: {}'.format(type(e).__name__, e))
	return out


def SLICES(n_cells):
	"""Digit planes of the integer Gram engine at this cell count under the current NRM_GRAM (0: the fp64 Gram kernel)."""
	from normalisr_amd.engine import Engine
	mode = os.environ.get('NRM_GRAM', 'i8')
	return {'i8': 6, 'i8x5': 5, 'f64': 0}[mode] if Engine.I8_MIN_CELLS <= n_cells < Engine.I8_MAX_CELLS else 0


def ARITH(n_cells):
	"""`dtype` of the JSON line: the arithmetic the dominant kernel computes in."""
	ns = SLICES(n_cells)
	return 'i8 digits x i8 -> i32 exact, {}-bit fixed point, f64 combine (f64 residuals, sums of squares and P-values)'.format(8 * ns - 2) if ns else 'f64'


def self_launch(args):
	"""`python bench.py --gpus N` outside a launcher: start N ranks with torch.distributed.run BEFORE this process
	touches the GPU (device_count does not initialise it) and hand their output through."""
	import socket
	import subprocess
	import torch
	have = torch.cuda.device_count()
	env = dict(os.environ)
	env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
	if have < args.gpus and env.get('NRM_SHARE_GPU') != '1':
		raise SystemExit('bench.py --gpus {}: only {} GPU(s) visible (NRM_SHARE_GPU=1 NRM_DIST_BACKEND=gloo runs the ranks on one GPU for a functional check)'.format(args.gpus, have))
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
		   '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
	return subprocess.run(cmd, env=env).returncode


class Ranks:
	"""The process group of this run (or a single process)."""

	def __init__(self, args):
		import torch
		self.torch = torch
		self.world = int(os.environ.get('WORLD_SIZE', '1'))
		self.rank = int(os.environ.get('RANK', '0'))
		local_rank = int(os.environ.get('LOCAL_RANK', '0'))
		self.backend = os.environ.get('NRM_DIST_BACKEND', 'nccl')  # 'gloo' + NRM_SHARE_GPU=1: functional test of the N>1 path on a 1-GPU box
		if os.environ.get('NRM_SHARE_GPU') == '1':
			local_rank = 0
		torch.cuda.set_device(local_rank)
		self.device = torch.device('cuda', local_rank)
		self.group = None
		self.ranks_seen = 1
		self.forced = os.environ.get('NRM_FORCE_EXCHANGE', '') if self.world == 1 else ''
		if self.forced:  # the N > 1 exchange code on one rank: an RCCL group of one (see CoexPlan, NRM_FORCE_EXCHANGE)
			import socket
			s = socket.socket()
			s.bind(('127.0.0.1', 0))
			os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
			os.environ.setdefault('MASTER_PORT', str(s.getsockname()[1]))
			s.close()
		if self.world > 1 or self.forced:
			import torch.distributed as dist
			if self.forced:
				dist.init_process_group('nccl', rank=0, world_size=1, device_id=self.device)
			elif self.backend == 'nccl':
				dist.init_process_group('nccl', device_id=self.device)
			else:
				dist.init_process_group(self.backend)
			self.group = dist.group.WORLD
			one = torch.ones(1, device=self.device if self.backend == 'nccl' else 'cpu', dtype=torch.float64)
			dist.all_reduce(one)  # how many ranks the collective library really connected
			self.ranks_seen = int(one.item())

	def barrier(self):
		if self.world > 1:
			self.torch.distributed.barrier()
		self.torch.cuda.synchronize()

	def max_over_ranks(self, v):
		if self.world == 1:
			return v
		t = self.torch.tensor([v], device=self.device if self.backend == 'nccl' else 'cpu', dtype=self.torch.float64)
		self.torch.distributed.all_reduce(t, op=self.torch.distributed.ReduceOp.MAX)
		return float(t.item())

	def close(self):
		if self.world > 1 or self.forced:
			self.torch.distributed.destroy_process_group()


def preflight_c5(rk, rows, cells):
	"""Per-rank HBM budget of the configs[4] workload, printed (stderr) and enforced before anything is allocated: input rows, K1's
	digit planes, the gather buffers of the exchange (every rank's planes), the dot blocks and results of the rank's block pairs.
	Results stay in HBM in the bench (no /dev/shm arrays; the sharded CLI checks those itself)."""
	torch = rk.torch
	world = rk.world
	rp = (rows + 127) // 128 * 128
	ns = SLICES(cells) or 8
	need = dict(input_rows=rows * cells * 8, digit_planes=rp * cells * ns, gather_buffers=(world * rp * cells * ns if world > 1 else 0),
				dot_and_results=(world // 2 + 1) * rp * rp * (8 + 2 * 8), generator_temporaries=2 * min(rows, 512) * cells * 8)
	total = sum(need.values())
	free, cap = torch.cuda.mem_get_info(rk.device)
	shm = None
	try:
		st = os.statvfs('/dev/shm')
		shm = st.f_bavail * st.f_frsize
	except OSError:
		pass
	msg = 'rank {}/{}: configs[4] workload needs {:.1f} GB of HBM ({}), {:.1f} GB free of {:.1f}; /dev/shm {} (not used by the bench: results stay in HBM)'.format(
		rk.rank, world, total / 1e9, ', '.join('{} {:.1f}'.format(k, v / 1e9) for k, v in need.items()), free / 1e9, cap / 1e9,
		'unknown' if shm is None else '{:.1f} GB free'.format(shm / 1e9))
	print(msg, file=sys.stderr, flush=True)
	if total > 0.95 * free:
		raise SystemExit('bench.py: ' + msg + ' -- does not fit; use --c5-rows / --c5-cells to shrink the per-rank block')


def timed_steps(rk, plan, steps, warmup, events_inside):
	"""W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
	for _ in range(warmup):
		plan.step()
	rk.barrier()
	t0 = time.perf_counter()
	for _ in range(steps):
		plan.step(timed=events_inside)
	rk.barrier()
	return rk.max_over_ranks(time.perf_counter() - t0)


def self_check(rk, plan, dt_local, dc, n_rows=12):
	"""The N > 1 line checks itself (round-5 verdict, item 5): after the timed region, untimed, the ranks pool a few RAW gene rows each (>= n_rows in all) and
	the P-values of those rows against each other AS THE N RANKS COMPUTED THEM (every unordered block pair was computed by exactly one rank: each rank
	contributes the entries of its own block pairs' outputs); then the SAME device computes those pairs as one rank would -- one coex call on the pooled rows,
	no exchange -- and the two must agree to 1e-6 relative (the north star's bar).  A pair's P-value depends on its two rows and the covariates alone, so
	the sub-problem's values are the full problem's.  Two small all-gathers (the collective the data path itself uses) and nothing point-to-point; the oracle
	is not involved (tests/ hold the device to it)."""
	import numpy as np
	import torch.distributed as dist
	from normalisr_amd.association import inv_rank
	torch, world, rank = rk.torch, rk.world, rk.rank
	dev = dt_local.device
	R = plan.rows
	m = max(1, -(-n_rows // world))
	idx_h = sorted(set(int(v) for v in np.linspace(0, R - 1, m).round()))
	m = len(idx_h)
	idx = torch.tensor(idx_h, device=dev)
	# what THIS rank computed of the pooled pairs: entry (bi m + a, bj m + b) = P[gene idx[a] of block bi, gene idx[b] of block bj]; NaN = not mine
	mine = torch.full((world * m, world * m), float('nan'), dtype=torch.float64, device=dev)
	as_t = lambda v: v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))
	for o in plan.outputs:
		p = as_t(o['p']).to(dev)
		sel = [(a, r - o['row_lo']) for a, r in enumerate(idx_h) if o['row_lo'] <= r < o['row_lo'] + o['nx']]
		if not sel:
			continue
		rows = torch.tensor([r for _, r in sel], device=dev)
		sub = p.index_select(0, rows).index_select(1, idx).to(torch.float64)  # (rows of block bi that are pooled, the m pooled columns of block bj)
		for (a, _), line in zip(sel, sub):
			mine[o['bi'] * m + a, o['bj'] * m:(o['bj'] + 1) * m] = line
	if os.environ.get('NRM_BENCH_FAULT') == 'p' and rank == world - 1:
		known = torch.nonzero(~torch.isnan(mine) & (mine > 0))
		off = known[known[:, 0] != known[:, 1]][0]
		mine[off[0], off[1]] *= 1.001  # (tests: one P-value of the last rank off by 1e-3 -- the run must fail)
	on_host = rk.backend != 'nccl'
	send = lambda t: t.contiguous().cpu() if on_host else t.contiguous()
	mine_rows = send(dt_local.index_select(0, idx))
	# (outputs as the CONCATENATION of the ranks' pieces along the first axis: the one shape both RCCL and gloo accept for this collective)
	rows_all = torch.empty((world * m, mine_rows.shape[1]), dtype=mine_rows.dtype, device=mine_rows.device)
	p_all = torch.empty((world * world * m, world * m), dtype=torch.float64, device=mine_rows.device)
	dist.all_gather_into_tensor(rows_all, mine_rows)
	dist.all_gather_into_tensor(p_all, send(mine))
	xs = rows_all.to(dev)
	p_all = p_all.reshape(world, world * m, world * m).to(dev)
	covered = (~torch.isnan(p_all)).sum(dim=0)  # how many ranks computed each ordered entry
	p_dist = torch.nan_to_num(p_all, nan=0.0).sum(dim=0)
	have = covered > 0
	p_dist = torch.where(have, p_dist / covered.clamp(min=1), p_dist.t() / covered.t().clamp(min=1))  # (i, j) from whoever computed (i, j) or (j, i)
	have = have | have.t()
	p_dist = p_dist.cpu().numpy()
	dc_h = dc.cpu().numpy().astype(np.float64)
	dci, dcr = inv_rank(dc_h @ dc_h.T)
	res = plan.be.eng.association_single0(xs, None, dc_h, dci, dcr, 0, True, False, np.float64 if dt_local.dtype == torch.float64 else np.float32)
	p_one = np.asarray(res['p'], dtype=np.float64)
	off = ~np.eye(world * m, dtype=bool)
	complete = bool(have.cpu().numpy()[off].all()) and int(covered.max()) <= 2  # every pooled pair computed by some rank (its diagonal-block entries by one rank twice: both orders)
	big = off & (p_one > 1e-290)
	rel = float(np.max(np.abs(p_dist[big] - p_one[big]) / p_one[big])) if big.any() else 0.0
	tiny_ok = bool((np.abs(p_dist[off & ~big] - p_one[off & ~big]) <= 1e-290).all())
	ok = bool(complete and rel <= 1e-6 and tiny_ok and rk.ranks_seen == world and np.isfinite(p_dist[off]).all())
	# one decision of all ranks
	t = torch.tensor([0.0 if ok else 1.0], device=rk.device if rk.backend == 'nccl' else 'cpu', dtype=torch.float64)
	dist.all_reduce(t)
	return dict(ok=bool(t.item() == 0), ranks_seen_by_collective=rk.ranks_seen, ranks=world, gene_rows_checked=world * m, pairs_checked=int(off.sum()) // 2,
				every_pair_computed_by_a_rank=complete, max_relative_p_difference=rel, tolerance=1e-6,
				against='the same device as ONE rank: coex on the pooled raw rows, no exchange (a pair\'s P-value depends on its two rows and the covariates alone)')


def guard_verdict(flags, eng):
	"""The integer engine's accuracy guard over every step that accumulated into `flags` (engine.new_flags): pairs whose P-value
	it could not certify to the tolerance (a user call would have been redone on the fp64 kernel) and the largest error estimate."""
	if flags is None or flags.shape[0] < 4:
		return None
	f = flags.cpu().numpy()
	return dict(uncertified_pairs=int(f[2]), largest_relative_p_error_bound=float(f[3:4].view(np.float32)[0]), tolerance=eng.guard_tol,
				note='rigorous bound (csrc/nrm_fix.h); 0 uncertified pairs = every P-value of the timed steps came from the integer engine')


def bench_coex(rk, nd, steps, warmup, rows_local, n, seed, dtype, label, loading=0.3):
	"""Sharded coex: every rank owns `rows_local` gene rows (generated on its GPU), see normalisr_amd.distributed.CoexPlan."""
	torch = rk.torch
	world, rank = rk.world, rk.rank
	ng = rows_local * world
	dt_local, dc = synth_c2(rows_local, n, seed, rk.device, torch, row0=rank * rows_local, dtype=dtype, loading=loading)
	plan = nd.CoexPlan(dt_local, dc, rank=rank, world=world, group=rk.group)
	# N = 1: the kernels are bracketed by HIP events inside the timed region (one stream, the events cost nothing).
	# N > 1: the exchange runs on RCCL's stream; timing events recorded on the launch stream were measured to slow
	# cross-stream work of the same process on ROCm 7.2 (see the end_to_end_pcie note), so the timed region runs without
	# them and the per-kernel breakdown comes from three extra steps after it (`kernels_ms_from`).
	events_inside = world == 1 and not plan.multi
	elapsed = timed_steps(rk, plan, steps, warmup, events_inside)
	if not events_inside:
		for _ in range(3):
			plan.step(timed=True)
		rk.barrier()
	tests = ng * (ng - 1) // 2
	gram_ms = plan.gram_ms()  # average duration of the dominant kernel launch(es) per step on this rank
	flops = 2.0 * n * plan.local_pair_count()  # algorithmic: 2 n_cell flop per test, tests this rank's launches cover
	esz = 4 if dtype == torch.float32 else 8
	out = dict(value=tests * steps / elapsed, unit='tests/s', steps=steps, warmup=warmup, ms_per_step=1e3 * elapsed / steps,
			   scaling='weak', dtype=ARITH(n),
			   config=dict(workload=label.format(genes=ng, cells=n), genes=ng, cells=n, covariates=3, tests_per_step=tests,
						   parallelism='gene-row blocks x{}'.format(world), exchange=None if not plan.multi else (
							   'all-gather of raw fp32 blocks' if plan.exchange_raw else
							   'fixed-point digit planes in {} cell chunks, one all-gather each, block pairs accumulated as the chunks land'.format(plan.chunks)
							   if plan.chunks else 'all-gather of residual blocks as fixed-point digit planes + exponents + sums of squares'),
						   exchange_bytes_per_rank=None if world == 1 else int((world - 1) * rows_local * n * (esz if plan.exchange_raw else (SLICES(n) or 8)))),
			   roofline=gram_roofline(n, flops, gram_ms, plan.rows_pad, plan.k_pad),
			   kernels_ms=plan.kernel_breakdown(), kernels_ms_from='timed region' if events_inside else '3 extra steps after the timed region',
			   guard=guard_verdict(plan.flags, plan.be.eng))
	out['kernels_roofline'] = side_rooflines(out['kernels_ms'], rows_local, n, esz, SLICES(n), plan.local_pair_count(), esz)
	if world > 1:
		out['self_check'] = self_check(rk, plan, dt_local, dc)
	return out, plan


def bench_de(rk, nd, steps, warmup, which, covariates=20):
	"""de shapes of BASELINE configs[2] (1 x 20k x 100k, 20 covariates: HBM-bound streaming path) and configs[3]
	(1k gRNAs x 15k genes x 50k cells: MFMA-bound general path); gene rows of Y sharded over the ranks, no collective."""
	torch = rk.torch
	world, rank, device = rk.world, rk.rank, rk.device
	if which == 'de_c3':
		nx, ny, n, nc, seed = 1, 20000, 100000, covariates, 3
	else:
		nx, ny, n, nc, seed = 1000, 15000, 50000, 5, 4
	ny_local = ny // world
	g = torch.Generator(device=device)
	g.manual_seed(seed)
	dc = torch.cat([torch.randn((nc - 1, n), generator=g, device=device, dtype=torch.float32), torch.ones((1, n), device=device, dtype=torch.float32)])
	p1 = 0.5 if nx == 1 else 0.01
	dx = (torch.rand((nx, n), generator=g, device=device) < p1).to(torch.float32)
	g2 = torch.Generator(device=device)
	g2.manual_seed(seed * 7919 + rank)
	dy = torch.randn((ny_local, n), generator=g2, device=device, dtype=torch.float32)
	dy[:16] += 0.2 * dx[0]
	plan = nd.DePlan(dx, dy, dc, rank=rank, world=world)
	elapsed = timed_steps(rk, plan, steps, warmup, True)
	tests = nx * ny_local * world
	ms = plan.step_ms()
	# per-kernel split of the step: three more steps, eager (a captured graph cannot be bracketed kernel by kernel), with the engine's
	# event trace on
	eng = plan.eng
	plan._graph.enabled, plan._graph.graph = False, None
	eng.trace = []
	for _ in range(3):
		plan.step()
	torch.cuda.synchronize()
	split = {}
	for name, e0, e1 in eng.trace:
		split[name] = split.get(name, 0.0) + e0.elapsed_time(e1) / 3
	eng.trace = None
	if plan.streaming():
		byts = 4.0 * n * ny_local  # algorithmic: every fp32 expression value read once
		roof = dict(bound='hbm', kernel='k_gram_skinny + sweep (whole step)', achieved=byts / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
					frac=byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None, kernel_ms=ms, pmc_kernels=['k_gram_skinny'])
	elif 'de_sparse' in split:
		# the design matrix is sparse (gRNA incidence, 1 % of the entries set): no K1 on the genes, no K2 -- the raw expression rows are read
		# (twice: sums with the covariates, then the gathers) and of each only the values at the design's entries are added up
		byts = 4.0 * n * ny_local  # algorithmic: every fp32 expression value read once
		kms = split['de_sparse'] + split.get('row_sums', 0.0)
		roof = dict(bound='hbm', kernel='k_de_sparse', achieved=byts / (kms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s', frac=byts / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
					traffic=None, algorithmic_bytes=byts, kernel_ms=kms, step_ms=ms, pmc_kernels=['k_de_sparse'] + (['k_s1_stream'] if 'row_sums' in split else []),
					note=('kernel_ms = k_s1_stream (sums of the rows with the covariates: a pass at HBM rate) + k_de_sparse' if 'row_sums' in split else 'kernel_ms = k_de_sparse (row sums with the covariates inside it: one pass over the rows)') +
						 ', which is bound by its vector ALU work (one fp32 -> fp64 conversion and one fp64 add per gathered value), not by HBM; NRM_DE_SPARSE=0 puts K1 + the integer Gram engine back',
					dense_path_flop_equivalent_tflops=2.0 * n * nx * ny_local / (kms * 1e-3) / 1e12)
		# A COLD call: a design tensor the engine has not seen (every step of the timed region reuses the lists the plan keeps for its
		# design; a one-shot norm.de pays for them).  Fresh copies of the design are made outside the timed region; every timed call builds
		# the lists (csrc/nrm_design_lists.hip), the design rows' statistics and everything else of a resident step, eagerly.
		fresh = [dx.clone() for _ in range(max(2, min(steps, 5)) + 1)]
		call = lambda d: eng.association_single0(d, dy, plan.dc64, plan.dci, plan.dcr, 0, return_dot=False, want_alpha=False, out_dtype=plan.out_dtype, cov=plan.cov, resident=True)
		call(fresh.pop())
		rk.barrier()
		t0 = time.perf_counter()
		for d in fresh:
			call(d)
		rk.barrier()
		cold_ms = 1e3 * rk.max_over_ranks(time.perf_counter() - t0) / len(fresh)
		eng.trace = []
		call(dx.clone())
		torch.cuda.synchronize()
		cold_split = {}
		for name, e0, e1 in eng.trace:
			cold_split[name] = round(cold_split.get(name, 0.0) + e0.elapsed_time(e1), 4)
		eng.trace = None
	else:
		roof = gram_roofline(n, 2.0 * n * nx * ny_local, split.get('gram', ms), 0, 0)  # K2 alone
		roof['algorithmic_bytes'] = float(SLICES(n) or 8) * (nx + ny_local) * n
		roof['step_ms'] = ms
	extra = {}
	if 'de_sparse' in split:
		extra = dict(cold_ms=cold_ms, cold_kernels_ms=cold_split, cold_note='a call on a design tensor the engine has not seen (lists built by csrc/nrm_design_lists.hip inside the call); '
					 'ms_per_step is a resident DePlan step, which keeps the lists of its design', cold_tests_per_s=tests / (cold_ms * 1e-3))
	return dict(extra, value=tests * steps / elapsed, unit='tests/s', steps=steps, warmup=warmup, ms_per_step=1e3 * elapsed / steps,
				scaling='strong', dtype='f64' if (plan.streaming() or 'de_sparse' in split) else ARITH(n),
				config=dict(workload='norm.de {} x {} genes x {} cells, fp32 input, {} covariates (BASELINE configs[{}])'.format(
					nx, ny, n, nc, 2 if which == 'de_c3' else 3), parallelism='gene rows of Y x{}, no collective'.format(world)), roofline=roof,
				kernels_ms={k: round(v, 4) for k, v in split.items()}, kernels_ms_from='3 extra eager steps after the timed region',
				guard=(dict(uncertified_pairs=0, note='fp64 sums over the design entries (csrc/nrm_de_sparse.hip): the integer engine and its guard are not involved')
					   if 'de_sparse' in split else guard_verdict(plan.result.get('flags'), eng)))


def bench_de_method(rk, steps, warmup, single):
	"""BASELINE configs[3] as the reference's CRISPR example runs it besides the naive test (cmd_highmoi.sh:19-22): norm.de with
	single=4 (`-m covariate`: every other gRNA a covariate, association.py:421-576,926-980) or single=1 (`-m single`: every gRNA on the
	cells free of the others, :263-390,911-925).  1000 gRNAs x 15000 genes x 50000 cells, inputs resident in HBM, results left there;
	a step is the whole call: A A^T and its rank certificate, K1 on design and genes, the large contraction on the integer engine,
	the host's 1000 x 1000 inverse (overlapped), B = G N, the sweep.  Gene rows sharded over the ranks, no collective."""
	torch = rk.torch
	from normalisr_amd.engine import get_engine
	world, rank, device = rk.world, rk.rank, rk.device
	nx, ny, n, nc, seed = 1000, 15000, 50000, 5, 4
	ny_local = ny // world
	g = torch.Generator(device=device)
	g.manual_seed(seed)
	dc = torch.cat([torch.randn((nc - 1, n), generator=g, device=device, dtype=torch.float32), torch.ones((1, n), device=device, dtype=torch.float32)])
	# single=1 tests a gRNA on the cells that carry no OTHER gRNA (association.py:915-918 asserts there are some): a low-MOI design, one
	# gRNA per cell on average; single=4 takes the high-MOI design of configs[3] (10 per cell)
	dx = (torch.rand((nx, n), generator=g, device=device) < (0.01 if single == 4 else 0.001)).to(torch.float32)
	g2 = torch.Generator(device=device)
	g2.manual_seed(seed * 7919 + rank)
	dy = torch.randn((ny_local, n), generator=g2, device=device, dtype=torch.float32)
	dy[:16] += 0.2 * dx[0]
	dc_h = dc.cpu().numpy().astype(np.float64)
	eng = get_engine(device.index)
	if single == 4:
		from normalisr_amd.single4 import association_tests_single4 as fn
	else:
		from normalisr_amd.single1 import association_tests_single1 as fn

	class Plan:
		out = None

		def step(self, timed=False):
			self.out = None  # (the previous step's results go back to the allocator first: no second set of GB-sized buffers)
			self.out = fn(dx, dy, dc_h, return_dot=False, device_out=True)

		def check(self):
			pass
	# a resident screen: the plan replays a step as one HIP graph; nothing of a step runs on the host (round 6)
	if single == 1:
		from normalisr_amd.single1 import Single1Plan
		plan = Single1Plan(dx, dy, dc_h, return_dot=False)
	elif os.environ.get('NRM_BENCH_S4', 'plan') == 'plan':
		from normalisr_amd.single4 import Single4Plan
		plan = Single4Plan(dx, dy, dc_h, return_dot=False)
	else:
		plan = Plan()
	for _ in range(3):  # the allocator's pool and the kernels' code objects settle in the first calls
		plan.step()
	elapsed = timed_steps(rk, plan, steps, warmup, False)
	plan.check()  # (the reference's assertions over the timed steps: device counters, looked at once)
	guard = dict(eng.last_guard)
	eng.trace = []
	for _ in range(2):
		plan.step()
	torch.cuda.synchronize()
	split = {}
	for name, e0, e1 in eng.trace:
		split[name] = split.get(name, 0.0) + e0.elapsed_time(e1) / 2
	eng.trace = None
	tests = nx * ny_local * world
	ms = 1e3 * elapsed / steps
	cold = {}
	if single == 1:
		# a COLD call: the public entry on a design tensor the engine has not seen -- entry lists, a plan's buffers, one step, the counters read
		fresh = [dx.clone() for _ in range(max(2, min(steps, 5)) + 1)]
		fn(fresh.pop(), dy, dc_h, return_dot=False, device_out=True)
		rk.barrier()
		t0 = time.perf_counter()
		for d in fresh:
			out = fn(d, dy, dc_h, return_dot=False, device_out=True)
		torch.cuda.synchronize()
		rk.barrier()
		cold_ms = 1e3 * rk.max_over_ranks(time.perf_counter() - t0) / len(fresh)
		del out, fresh
		cold = dict(cold_ms=cold_ms, cold_tests_per_s=tests / (cold_ms * 1e-3), cold_note='association_tests(single=1) on a design tensor the engine has not seen: entry lists, buffers, one step, the counters read back; ms_per_step is a resident Single1Plan step (one HIP graph)')
	if single == 4:
		# a COLD call: a design tensor the engine has not seen (the timed steps reuse the lists kept for their design tensor; single=1 lists
		# its design anew in every call).  Fresh copies are made outside the timed region.
		fresh = [dx.clone() for _ in range(max(2, min(steps, 5)) + 1)]
		fn(fresh.pop(), dy, dc_h, return_dot=False, device_out=True)
		rk.barrier()
		t0 = time.perf_counter()
		for d in fresh:
			plan.out = None
			plan.out = fn(d, dy, dc_h, return_dot=False, device_out=True)
		rk.barrier()
		cold_ms = 1e3 * rk.max_over_ranks(time.perf_counter() - t0) / len(fresh)
		cold = dict(cold_ms=cold_ms, cold_tests_per_s=tests / (cold_ms * 1e-3), cold_note='a call on a design tensor the engine has not seen (its lists built inside the call); ms_per_step reuses the lists kept for the design tensor')
	if single == 4 and 'de_sparse' in split:  # the design is sparse: Y~ X~^T from the raw expression rows at the design's entries (csrc/nrm_de_sparse.hip)
		byts = 4.0 * n * ny_local
		kms = split['de_sparse'] + split.get('row_sums', 0.0)
		roof = dict(bound='hbm', kernel='k_de_sparse', achieved=byts / (kms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s', frac=byts / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
					traffic=None, algorithmic_bytes=byts, kernel_ms=kms, step_ms=ms, pmc_kernels=['k_de_sparse'] + (['k_s1_stream'] if 'row_sums' in split else []),
					note='kernel_ms = k_s1_stream (row sums with the covariates) + k_de_sparse (bound by its vector ALU work: a conversion and an fp64 add per gathered '
						 'value); the rest of a step: K1 on the design, M~ = X~ X~^T and its Newton-Schulz inverse on the fp64 matrix cores, B = G N~, the sweep')
		dtype = 'f64'
	elif single == 4:
		roof = gram_roofline(n, 2.0 * n * nx * ny_local, split.get('gram_yx', ms), 0, 0)  # the large contraction Y~ X~^T alone
		roof['algorithmic_bytes'] = float(SLICES(n) or 8) * (nx + ny_local) * n
		roof['step_ms'] = ms
		dtype = ARITH(n)
	else:
		kept = plan.cells_kept()
		byts = 4.0 * ny_local * (n + kept)  # every fp32 expression value read once + the values at the cells that carry one gRNA written once (transposed)
		kms = split.get('s1_stream', ms)
		roof = dict(bound='hbm', kernel='k_s1_stream', achieved=byts / (kms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
					frac=byts / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None, algorithmic_bytes=byts, kernel_ms=kms, step_ms=ms, cells_with_one_grna=kept,
					note='a step is seven launches replayed as one HIP graph: cell selection, the gRNAs\' own sums, their pseudo-inverses / ranks / P-value plans (a lane per gRNA: association.py:350-374), the stream kernel, the sweep -- nothing on the host')
		dtype = 'f64'
	return dict(cold, value=tests * steps / elapsed, unit='tests/s', steps=steps, warmup=warmup, ms_per_step=ms, scaling='strong', dtype=dtype,
				config=dict(workload='norm.de(single={}) {} gRNAs x {} genes x {} cells, fp32 input, {} covariates (BASELINE configs[3] as `normalisr de -m {}`, examples/GSE120861/code/cmd_highmoi.sh)'.format(
					single, nx, ny, n, nc, 'covariate' if single == 4 else 'single') + ('' if single == 4 else '; gRNA incidence 0.1 % (low MOI: single=1 needs cells with one gRNA)'), parallelism='gene rows of Y x{}, no collective'.format(world)),
				roofline=roof, kernels_ms={k: round(v, 4) for k, v in split.items()}, kernels_ms_from='2 extra steps after the timed region, eager, HIP events around the engine\'s launches (the timed steps are one HIP graph each)',
				guard=dict(uncertified_pairs=int(guard.get('hits', 0)), largest_relative_p_error_bound=float(guard.get('worst', 0.0)), tolerance=eng.guard_tol,
						   fp64_rerun=bool(guard.get('fallback', False))), metric='association tests/sec (de)')


def bench_binnet(rk, steps, warmup):
	"""binnet (binnet.py:77-173) on a 30 000 x 30 000 fp64 P-value matrix resident in HBM -- the consumer of configs[4]'s output
	(examples/GSE123139/code/cmd_coex.sh:40-42): per-row Benjamini-Hochberg threshold, one byte out per pair."""
	torch = rk.torch
	from normalisr_amd import binnet as nb
	ng = 30000
	if torch.cuda.mem_get_info(rk.device)[0] < 12 * (1 << 30):
		return None
	g = torch.Generator(device=rk.device)
	g.manual_seed(7)
	p = torch.rand((ng, ng), generator=g, device=rk.device, dtype=torch.float64)
	p[:, :64] *= 1e-6  # some strong columns so that rows have something to keep
	p = torch.triu(p, 1)
	p = p + p.T

	class Plan:
		def step(self, timed=False):
			self.out = nb.binnet(p, 0.05)
	plan = Plan()
	elapsed = timed_steps(rk, plan, steps, warmup, False)
	ms = 1e3 * elapsed / steps
	byts = float(ng) * ng * (8 + 1)
	kept = int(plan.out.sum())
	return dict(metric='binnet entries/sec', value=float(ng) * ng * steps / elapsed, unit='entries/s', steps=steps, warmup=warmup, ms_per_step=ms, scaling='single GPU',
				dtype='f64 compare / count', config=dict(workload='binnet on a {0} x {0} fp64 P-value matrix, qcut 0.05 (the output of BASELINE configs[4])'.format(ng), edges_kept=kept),
				roofline=dict(bound='hbm', kernel='k_binnet_rows', achieved=byts / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s', frac=byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
							  algorithmic_bytes=byts, traffic=None, kernel_ms=ms))


def bench_normvar(rk, steps, warmup):
	"""norm.normvar (norm.py:166-289) on a configs[1]-sized matrix, the step in front of the hot path: 5000 genes x 10 000 cells fp32,
	5 covariates, the matrix resident in HBM and the result left there (device_out=True: what coex / de take next) -- per-gene moments in one
	pass, a thread per gene solves its small OLS, one pass writes the result (csrc/nrm_normvar.hip).  The numpy -> numpy call (0.2 GB up,
	0.4 GB down over PCIe) is timed beside it."""
	torch = rk.torch
	import normalisr_amd.normalisr as norm
	from normalisr_amd.engine import get_engine
	eng = get_engine(rk.device.index)
	ng, n, nc = 5000, 10000, 5
	rng = np.random.default_rng(1)
	dt = rng.standard_normal((ng, n), dtype=np.float32) - 9
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
	w, wt = np.exp(0.25 * rng.normal(size=n)), rng.uniform(0, 1.5, ng)
	d_dt = torch.from_numpy(dt).to(rk.device)

	from normalisr_amd.norm import NormvarPlan
	plan = NormvarPlan(d_dt, dc, w, wt)  # (round 6: the host's share of a call -- log w, uploads, allocations, the flags' read-back -- done once; a step is the three kernels, one HIP graph)
	plan.step()
	elapsed = timed_steps(rk, plan, steps, warmup, False)
	plan.check()
	ms = 1e3 * elapsed / steps
	t0 = time.perf_counter()
	for _ in range(5):
		pub = norm.normvar(d_dt, dc, w, wt, device_out=True)
	torch.cuda.synchronize()
	public_ms = 1e3 * (time.perf_counter() - t0) / 5
	assert torch.equal(pub[0], plan.out)  # (the same kernels on the same inputs)
	del pub
	eng.trace = []
	norm.normvar(d_dt, dc, w, wt, device_out=True)
	torch.cuda.synchronize()
	split = {}
	for name, e0, e1 in eng.trace:
		split[name] = round(split.get(name, 0.0) + e0.elapsed_time(e1), 4)
	eng.trace = None
	t0 = time.perf_counter()
	norm.normvar(dt, dc, w, wt)
	host_ms = 1e3 * (time.perf_counter() - t0)
	byts = float(ng) * n * (4 + 8)  # algorithmic: the fp32 matrix read once, the fp64 result written once
	kms = sum(split.values()) or ms
	return dict(metric='normvar values/sec (resident)', value=float(ng) * n * steps / elapsed, unit='values/s', steps=steps, warmup=warmup, ms_per_step=ms,
				scaling='single GPU', dtype='f64', config=dict(workload='norm.normvar {} genes x {} cells fp32, {} covariates, matrix resident in HBM, result left there'.format(ng, n, nc)),
				kernels_ms=split, numpy_in_out_ms=host_ms, public_call_resident_ms=public_ms,
				roofline=dict(bound='hbm', kernel='k_nv_moments + k_nv_solve + k_nv_apply', achieved=byts / (kms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s', frac=byts / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
							  algorithmic_bytes=byts, traffic=None, kernel_ms=kms, step_ms=ms, pmc_kernels=['k_nv_moments', 'k_nv_apply'],
							  note='the matrix is read twice (moments, then the result); the result pass runs at the HBM rate, the moments pass re-reads the covariates per gene from L2 (DESIGN.md section 6); a step also uploads the covariates and weights (0.5 MB) and reads one word of flags back'))


def bench_chain(rk, steps, warmup):
	"""The reference's pipeline normvar -> coex -> binnet (examples/GSE123139/code/cmd_coex.sh:38-46; norm.py:166-289 feeds coex.py:46-48 feeds
	binnet.py:134-173) at configs[1] size as ONE resident chain: the expression matrix, the normalised matrix, the P-values and the network stay in
	HBM between the three steps (the reference passes them through files)."""
	torch = rk.torch
	import normalisr_amd.normalisr as norm
	from normalisr_amd.binnet import binnet
	from normalisr_amd.engine import get_engine
	eng = get_engine(rk.device.index)
	ng, n = 5000, 10000
	d_dt, dc = synth_c2(ng, n, 2, rk.device, torch)
	d_dt -= 9
	dc = dc.cpu().numpy().astype(np.float64)
	rng = np.random.default_rng(1)
	w, wt = np.exp(0.25 * rng.normal(size=n)), rng.uniform(0, 1.5, ng)

	from normalisr_amd.norm import NormvarPlan
	nv = NormvarPlan(d_dt, dc, w, wt)

	class Plan:
		def step(self, timed=False):
			self.net = None
			dtn = nv.step()
			p, dot, var = norm.coex(dtn, nv.dcn, device_out=True)
			self.net = binnet(p, 0.05)
	plan = Plan()
	plan.step()
	elapsed = timed_steps(rk, plan, steps, warmup, False)
	ms = 1e3 * elapsed / steps
	eng.trace = []
	plan.step()
	torch.cuda.synchronize()
	split = {}
	for name, e0, e1 in eng.trace:
		split[name] = round(split.get(name, 0.0) + e0.elapsed_time(e1), 4)
	eng.trace = None
	tests = ng * (ng - 1) // 2
	roof = gram_roofline(n, 2.0 * n * (tests + ng), split.get('gram', ms), 5120, n)
	roof['step_ms'] = ms
	return dict(metric='association tests/sec (normvar -> coex -> binnet, resident)', value=tests * steps / elapsed, unit='tests/s', steps=steps, warmup=warmup, ms_per_step=ms,
				scaling='single GPU', dtype=ARITH(n), config=dict(workload='norm.normvar -> norm.coex -> binnet, {} genes x {} cells fp32, 3 covariates, nothing leaves HBM between the steps'.format(ng, n),
																 edges_kept=int(plan.net.sum())), kernels_ms=split, roofline=roof)


def bench_c5_full(rk):
	"""BASELINE configs[4] WHOLE on one GPU: 30 000 genes x 500 000 cells fp64 (4.5e8 pairs).  The 120 GB matrix is generated and
	dropped in blocks of 3840 gene rows; only its 90 GB of digit planes stay resident (engine.coex_blocks_resident).  value = pairs /
	(K1 + K2 + K3 time, HIP events); generating the synthetic blocks between the K1 launches is not part of the hot path."""
	torch = rk.torch
	from normalisr_amd.association import _prepare_covariates
	from normalisr_amd.engine import get_engine
	eng = get_engine(rk.device.index)
	ng, n = 30000, C5_CELLS
	if torch.cuda.mem_get_info(rk.device)[1] < 200 * (1 << 30):
		return None
	rows = lambda lo, hi: synth_c2(hi - lo, n, 5, rk.device, torch, row0=lo, dtype=torch.float64, loading=0.05)[0]
	dc = synth_c2(1, n, 5, rk.device, torch, dtype=torch.float64)[1].cpu().numpy()
	dc64, dci, dcr = _prepare_covariates(dc)
	eng.trace = []
	rk.barrier()
	t0 = time.perf_counter()
	res = eng.coex_blocks_resident(rows, ng, n, dc64, dci, dcr, 0, np.float64)
	rk.barrier()
	wall = time.perf_counter() - t0
	split = {}
	for name, e0, e1 in eng.trace:
		split[name] = split.get(name, 0.0) + e0.elapsed_time(e1)
	eng.trace = None
	guard = guard_verdict(res['flags'], eng)
	del res
	torch.cuda.empty_cache()
	tests = ng * (ng - 1) // 2
	ms = sum(split.values())
	roof = gram_roofline(n, 2.0 * n * (tests + ng), split['gram'], 30080, 500000)
	roof['note'] = 'sum of the {} band launches of k_gram_i8'.format(len([1 for t in range(0, ng, 1024)]))
	return dict(metric='association tests/sec (gene x gene coex)', value=tests / (ms * 1e-3), unit='tests/s', steps=1, warmup=0, ms_per_step=ms,
				scaling='single GPU', dtype=ARITH(n),
				config=dict(workload='norm.coex gene x gene, 30000 genes x 500000 cells, fp64 input, 3 covariates (BASELINE configs[4], the WHOLE problem on one GPU)',
							genes=ng, cells=n, tests_per_step=tests, parallelism='one GPU; input generated in blocks of 3840 rows, 90 GB of digit planes resident'),
				roofline=roof, kernels_ms={k: round(v, 2) for k, v in split.items()}, wall_seconds_incl_generating_the_input=round(wall, 2), guard=guard)


def short(v, n=160):
	return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + '...'

def contract_line(head, extras, world, e2e=None):
	"""The one line the driver reads: BASELINE's metric on its config, roofline, cpu_baseline, the scaling series, and one summary row
	per extra workload -- kept under 4 KB (the full records are the lines before it)."""
	roof = {k: short(v, 120) for k, v in head['roofline'].items() if k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes', 'kernel_ms',
																		  'frac_of_fp32_mfma_peak', 'frac_of_fp64_mfma_peak', 'effective_clock_ghz_profiled', 'traffic_source')}
	cb = head.get('cpu_baseline')
	line = dict(metric=head['metric'], value=head['value'], unit=head['unit'], n_gpus=world, steps=head['steps'], warmup=head['warmup'], ms_per_step=head['ms_per_step'],
				higher_is_better=True, scaling=head['scaling'], vs_baseline=None, dtype=short(head['dtype'], 90), data='synthetic',
				config={k: short(v) for k, v in head['config'].items()}, roofline=roof,
				cpu_baseline=None if not cb else {k: short(v, 200) for k, v in cb.items()}, scaling_series=head.get('scaling_series'),
				guard=None if not head.get('guard') else {k: v for k, v in head['guard'].items() if k != 'note'},
				kernels_ms={k: round(v, 4) for k, v in (head.get('kernels_ms') or {}).items()}, ranks_seen_by_collective=head['ranks_seen_by_collective'],
				dist_backend=head['dist_backend'], frac_of_fp32_mfma_peak=head.get('frac_of_fp32_mfma_peak'),
				end_to_end_pcie_s=None if not e2e else e2e['seconds'])
	if head.get('self_check') is not None:
		line['self_check'] = {k: v for k, v in head['self_check'].items() if k != 'against'}
	rows = {}
	for w, r in extras.items():
		if not isinstance(r, dict) or 'value' not in r:
			rows[w] = short(str(r.get('error', r)) if isinstance(r, dict) else str(r), 100)
			continue
		row = dict(value=float('%.4g' % r['value']), ms=round(r['ms_per_step'], 3), kernel=short(r['roofline']['kernel'], 24), frac=round(r['roofline']['frac'], 3))
		if 'cold_ms' in r:
			row['cold_ms'] = round(r['cold_ms'], 3)
		rows[w] = row
	line['extra_workloads'] = rows
	line['detail'] = 'full records: the JSON lines before this one (one per workload; also gpurun_out/bench_detail_n{}.jsonl)'.format(world)
	if '_abandoned' in extras:
		line['extras_abandoned'] = extras['_abandoned']
	return line



def main():
	ap = argparse.ArgumentParser()
	ap.add_argument('--gpus', type=int, default=1)
	ap.add_argument('--steps', type=int, default=20)
	ap.add_argument('--warmup', type=int, default=3)
	ap.add_argument('--genes', type=int, default=5000, help='genes at N=1 (scaled by sqrt(N) for N>1)')
	ap.add_argument('--cells', type=int, default=10000)
	ap.add_argument('--cpu-seconds', type=float, default=10.0, help='minimum CPU-baseline time (0 = skip)')
	ap.add_argument('--cpu-worker', nargs=6, default=None, help=argparse.SUPPRESS)
	ap.add_argument('--workload', default=None, choices=['coex_c2', 'de_c3', 'de_c4', 'coex_c5', 'coex_c5_f64', 'coex_c5_full_1gpu', 'de_c4_single4', 'de_c4_single1', 'coex_c2_f64', 'binnet_c5', 'normvar_c2', 'chain_c2'],
					help='headline workload.  Default: coex_c2 = BASELINE configs[1] at N=1; coex_c5 = configs[4] (3750 gene rows per rank x 500k cells) at N>1, '
					'the configuration the 8-GPU target is quoted on.  de_c3 / de_c4 = configs[2] / [3]')
	ap.add_argument('--c5-rows', type=int, default=C5_ROWS_PER_RANK, help='gene rows per rank of the coex_c5 workload (smaller: functional runs)')
	ap.add_argument('--c5-cells', type=int, default=C5_CELLS, help='cells of the coex_c5 workload (smaller: functional runs)')
	ap.add_argument('--no-extras', action='store_true', help='skip the extra_workloads entries (the other BASELINE configs)')
	ap.add_argument('--extras', default='', help='comma-separated subset of the extra workloads to run (default: all)')
	ap.add_argument('--extras-steps', type=int, default=5)
	ap.add_argument('--extras-timeout', type=float, default=480.0, help='seconds after which a stuck extra workload is abandoned and the headline line printed')
	ap.add_argument('--covariates', type=int, default=20, help='covariates of the de_c3 workload (<= 15 selects the half-width streaming kernel)')
	ap.add_argument('--seed', type=int, default=2)
	ap.add_argument('--e2e', type=int, default=2, help='repetitions of the numpy-in/numpy-out end-to-end timing (0 = skip)')
	args = ap.parse_args()
	if args.cpu_worker:
		w = args.cpu_worker
		cpu_baseline_worker(int(w[0]), int(w[1]), int(w[2]), int(w[3]), float(w[4]), bool(int(w[5])))
		return 0

	if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
		return self_launch(args)
	world = int(os.environ.get('WORLD_SIZE', '1'))
	if args.gpus != world:
		raise SystemExit('bench.py --gpus {} inside a launcher with WORLD_SIZE={}'.format(args.gpus, world))
	if args.workload is None:
		args.workload = 'coex_c2' if world == 1 else 'coex_c5'
	cpu = None
	if world == 1 and args.cpu_seconds > 0 and args.workload == 'coex_c2':
		# CPU baseline first, in a child process, before this process touches the GPU
		cpu = cpu_baseline(int(round(args.genes)), args.cells, 3, args.seed, args.cpu_seconds, extras=not args.no_extras)

	import torch
	from normalisr_amd import distributed as nd
	rk = Ranks(args)
	rank = rk.rank
	n = args.cells

	e2e = None
	if world == 1 and args.e2e > 0 and args.workload == 'coex_c2':
		# numpy in -> numpy out through the drop-in API (H2D + kernels + D2H over PCIe); reported beside `value`, never as it
		import normalisr_amd.normalisr as norm
		# (measured before any timing event exists in this process: after hipEvents with timing have been recorded,
		#  cross-stream copies of the same process run several times slower on ROCm 7.2 -- a bench artefact, not an API cost)
		dt_local, dc = synth_c2(args.genes, n, args.seed, rk.device, torch)
		h_dt, h_dc = dt_local.cpu().numpy(), dc.cpu().numpy()
		del dt_local, dc
		norm.coex(h_dt[:256], h_dc)
		ts = []
		for _ in range(args.e2e):
			res = None  # the previous results are released outside the timed region
			t1 = time.perf_counter()
			res = norm.coex(h_dt, h_dc)
			ts.append(time.perf_counter() - t1)
		e2e = dict(seconds=min(ts), all_seconds=[round(t, 5) for t in ts], tests_per_s=args.genes * (args.genes - 1) // 2 / min(ts),
				   note='norm.coex(numpy fp32) -> numpy, pageable host memory, PCIe inclusive')
		res = h_dt = None

	def run(which, steps, warmup):
		if which == 'coex_c2':
			# weak scaling: pairs per GPU fixed -> genes ~ sqrt(N); rounded so every rank owns the same number of rows
			rows_local = int(round(args.genes * np.sqrt(world) / world))
			out, plan = bench_coex(rk, nd, steps, warmup, rows_local, n, args.seed, torch.float32,
								   'norm.coex gene x gene, {genes} genes x {cells} cells, fp32 input, 3 covariates (BASELINE configs[1]' + ('' if world == 1 else ', genes scaled by sqrt(N)') + ')')
			out['metric'] = 'association tests/sec (gene x gene coex)'
			if world == 1 and rows_local == 5000 and n == 10000:
				pmc_traffic('coex_c2', out['roofline'])
			return out
		if which == 'coex_c5':
			preflight_c5(rk, args.c5_rows, args.c5_cells)
			out, plan = bench_coex(rk, nd, steps, warmup, args.c5_rows, args.c5_cells, 5, torch.float64,
								   'norm.coex gene x gene, {genes} genes x {cells} cells, fp64 input, 3 covariates (BASELINE configs[4] at %d gene rows per rank' % args.c5_rows +
								   ('; N=8 is the full 30k x 30k problem)' if world != 8 or args.c5_rows != C5_ROWS_PER_RANK else ': the full problem)'), loading=0.05)
			out['metric'] = 'association tests/sec (gene x gene coex)'
			out['scaling'] = 'weak'  # (the contract's word for a per-rank input that stays fixed as N grows; what grows with N is said beside it)
			out['scaling_note'] = 'gene rows per rank fixed; the block pairs a rank computes grow as (N+1)/2, the problem as N^2'
			if world == 1:
				pmc_traffic('coex_c5', out['roofline'])
			return out
		if which == 'coex_c5_full_1gpu':
			return bench_c5_full(rk)
		if which in ('de_c4_single4', 'de_c4_single1'):
			out = bench_de_method(rk, steps, warmup, 4 if which.endswith('4') else 1)
			kern = out['roofline'].pop('pmc_kernels', None)
			if world == 1:
				pmc_traffic(which + ('_dense' if out['roofline']['kernel'].startswith('k_gram') else ''), out['roofline'], kernels=kern)
			return out
		if which == 'normvar_c2':
			out = bench_normvar(rk, steps, warmup)
			pmc_traffic(which, out['roofline'], kernels=out['roofline'].pop('pmc_kernels', None))
			return out
		if which == 'chain_c2':
			return bench_chain(rk, steps, warmup)
		if which == 'binnet_c5':
			out = bench_binnet(rk, steps, warmup)
			if out is not None:
				pmc_traffic(which, out['roofline'], kernels=['k_binnet_rows'])
			return out
		if which in ('coex_c2_f64', 'coex_c5_f64'):  # configs[1] / the configs[4] slice on the fp64 matrix cores: the dtype the north star names literally
			prev = os.environ.get('NRM_GRAM')
			os.environ['NRM_GRAM'] = 'f64'
			try:
				out = run(which[:-4], steps, warmup)
			finally:
				if prev is None:
					del os.environ['NRM_GRAM']
				else:
					os.environ['NRM_GRAM'] = prev
			for k in ('traffic', 'traffic_source', 'traffic_unit', 'effective_clock_ghz_profiled'):  # (the counters on file are the integer engine's)
				out['roofline'].pop(k, None)
			out['roofline']['traffic'] = None
			return out
		out = bench_de(rk, nd, steps, warmup, which, args.covariates)
		out['metric'] = 'association tests/sec (de)'
		if world == 1:
			dense = out['roofline']['kernel'].startswith('k_gram_i8') or out['roofline']['kernel'].startswith('k_gram_f64')
			pmc_traffic(which + ('_dense' if dense else ''), out['roofline'], kernels=out['roofline'].pop('pmc_kernels', None))
		else:
			out['roofline'].pop('pmc_kernels', None)
		return out

	head = run(args.workload, args.steps, args.warmup)
	head = dict(dict(metric=head.pop('metric'), value=head.pop('value'), unit=head.pop('unit'), n_gpus=world, steps=head.pop('steps'),
					 warmup=head.pop('warmup'), ms_per_step=head.pop('ms_per_step'), higher_is_better=True, scaling=head.pop('scaling'),
					 vs_baseline=None, dtype=head.pop('dtype'), data='synthetic'), **head)
	head['ranks_seen_by_collective'] = rk.ranks_seen
	head['dist_backend'] = None if (world == 1 and not rk.forced) else ('rccl (torch nccl backend)' if (rk.backend == 'nccl' or rk.forced) else rk.backend)
	if rk.forced:
		head['forced_exchange'] = 'NRM_FORCE_EXCHANGE={}: the N > 1 exchange path on one rank (RCCL group of one; the own block pair is contracted from the gather buffers)'.format(rk.forced)
	cpu_extras = cpu.pop('extras', {}) if cpu else {}
	head['cpu_baseline'] = cpu
	head['end_to_end_pcie'] = e2e
	head['frac_of_fp32_mfma_peak'] = head['roofline'].get('frac_of_fp32_mfma_peak')  # the roofline BASELINE.json's north_star names

	# ONE workload is measured at every N: the per-rank slice of BASELINE configs[4] (3750 gene rows per rank x 500 000 cells fp64; N = 8 is
	# the full 30 000-gene problem the 8-GPU target is quoted on).  It is the headline at N > 1 and an extra at N = 1, where the headline is
	# configs[1]; `scaling_series` carries it under the same key on every line, so that a 1 -> 8 curve is read from one key.
	def series_of(r):
		return dict(workload='coex_c5: BASELINE configs[4] per-rank slice, {} gene rows per rank x {} cells fp64'.format(args.c5_rows, args.c5_cells), value=r['value'], unit=r['unit'],
					ms_per_step=r['ms_per_step'], ranks=world, tests_per_step=r['config']['tests_per_step'], value_per_rank=r['value'] / world,
					roofline_frac=r['roofline']['frac'], exchange_ms_not_hidden=r.get('kernels_ms', {}).get('exchange'))
	head['scaling_series'] = series_of(head) if args.workload == 'coex_c5' else None
	head['config']['scaling_series'] = 'top-level `scaling_series` = BASELINE configs[4] per-rank slice, the same workload at every N (the headline itself at N > 1)'

	printed = threading.Event()
	extras = {}
	detail_path = os.path.join(ROOT, 'gpurun_out', 'bench_detail_n{}.jsonl'.format(world))

	def say(obj):
		"""A JSON line that is NOT the contract line: the full record of an extra workload (or of the headline), printed as soon as it exists
		and appended to gpurun_out/bench_detail_nN.jsonl; the contract line, last and short, carries their summary."""
		if rank != 0:
			return
		line = json.dumps(obj)
		print(line, flush=True)
		try:
			os.makedirs(os.path.dirname(detail_path), exist_ok=True)
			with open(detail_path, 'a') as fh:
				fh.write(line + '\n')
		except OSError:
			pass

	def emit():
		if rank == 0 and not printed.is_set():
			printed.set()
			try:  # RCCL's version banner sits in the C library's stdout buffer until exit: out with it first, so that the JSON line is the last line
				import ctypes
				ctypes.CDLL(None).fflush(None)
			except Exception:  # noqa: BLE001
				pass
			print(json.dumps(contract_line(head, extras, world, e2e)), flush=True)

	if rank == 0:
		try:
			os.remove(detail_path)
		except OSError:
			pass
	say(dict(workload_detail=args.workload, **head))
	if not args.no_extras:
		# the other BASELINE configs, measured like the headline; a stuck collective must not lose the headline line
		def give_up():
			extras['_abandoned'] = 'extra workloads exceeded --extras-timeout {} s'.format(args.extras_timeout)
			emit()
			os._exit(0)
		dog = threading.Timer(args.extras_timeout, give_up)
		dog.daemon = True
		dog.start()
		names = [w for w in ('coex_c5', 'coex_c2', 'de_c3', 'de_c4', 'de_c4_single4', 'de_c4_single1', 'coex_c2_f64', 'coex_c5_f64') if w != args.workload and not (w == 'coex_c2' and world == 1)] + (
			['binnet_c5', 'normvar_c2', 'chain_c2', 'coex_c5_full_1gpu'] if world == 1 and not rk.forced else [])
		if world > 1:
			names = [w for w in names if w not in ('coex_c2_f64', 'coex_c5_f64')]
		if args.extras:
			names = [w for w in names if w in args.extras.split(',')]
		for w in names:
			try:
				torch.cuda.empty_cache()
				r = run(w, args.steps if w == 'de_c3' else args.extras_steps, 3 if w == 'de_c3' else 2)  # (the 2 ms step: enough of them to time)
				if r is None:
					continue
				r['n_gpus'] = world
				if w in cpu_extras or w.startswith('coex_c5'):
					r['cpu_baseline'] = cpu_extras.get(w, cpu_extras.get('coex_c5'))
				elif '_error' in cpu_extras:
					r['cpu_baseline'] = dict(error=cpu_extras['_error'])
				extras[w] = r
				if w == 'coex_c5':
					head['scaling_series'] = series_of(r)
				say(dict(workload_detail=w, **r))
			except Exception as e:  # reported, never fatal for the headline
				extras[w] = dict(error='{}: {}'.format(type(e).__name__, e))
				say(dict(workload_detail=w, **extras[w]))
		dog.cancel()
	emit()
	# N > 1: a line whose ranks did not all take part, or whose N-rank P-values differ from what this device computes as one rank, is a FAILED run
	bad = world > 1 and (rk.ranks_seen != world or not (head.get('self_check') or {}).get('ok', head.get('self_check') is None))
	if bad and rank == 0:
		print('bench.py: the N = {} run failed its own check: ranks seen by the collective {}, self_check {}'.format(world, rk.ranks_seen, head.get('self_check')), file=sys.stderr, flush=True)
	rk.close()
	return 3 if bad else 0


if __name__ == '__main__':
	sys.exit(main())
