#!/usr/bin/env python3
"""bench.py -- association tests/sec on the BASELINE.json workload.

N=1 workload = BASELINE.json configs[1]: norm.coex gene x gene on 5k genes x 10k cells, fp32 input,
3 covariates (2 random + intercept), seeded synthetic data.  A step is one full
pass of the hot path over the matrix resident in HBM: residualise (K1) -> fp64-MFMA Gram (K2) ->
per-pair sweep R^2 -> p, covariance (K3); outputs stay in HBM.  tests = ng(ng-1)/2 unique pairs.
For N>1 (one process per GPU, RCCL) the gene count grows as sqrt(N) so the pairs per GPU stay fixed
(weak scaling); gene-row blocks are residualised locally and exchanged by all-gather.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_gram_f64), timed live with HIP
events on the launch stream; `cpu_baseline` times the CPU oracle (a port of the reference's algorithm,
test infrastructure) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

PMC_FILE = os.path.join(ROOT, 'profiles', 'r01_pmc_c2.json')  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
F64_MFMA_PEAK_TFLOPS = 78.6  # v_mfma_f64_16x16x4_f64: 32 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz (= 1/2 of the 157.3 TF fp32 matrix peak of MI355X_MICROARCH.md)


def synth_c2(ng, n, seed, device, torch, row0=0):
	"""N(0,1) + 0.3 * loading * shared latent factor; dc = [2 x N(0,1); ones]; fp32."""
	g = torch.Generator(device=device)
	g.manual_seed(seed)
	lat = torch.randn((1, n), generator=g, device=device, dtype=torch.float32)
	dc = torch.cat([torch.randn((2, n), generator=g, device=device, dtype=torch.float32),
					torch.ones((1, n), device=device, dtype=torch.float32)])
	g2 = torch.Generator(device=device)
	g2.manual_seed(seed * 1000003 + row0)
	load = torch.randn((ng, 1), generator=g2, device=device, dtype=torch.float32)
	dt = torch.randn((ng, n), generator=g2, device=device, dtype=torch.float32) + 0.3 * load * lat
	return dt, dc


def cpu_baseline_worker(ng, n_cells, nc, seed, min_seconds):
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
		oracle.coex(dt, dc, nth=cores)
		reps += 1
		el = time.perf_counter() - t0
		if el >= min_seconds or reps >= 50:
			break
	pairs = ng * (ng - 1) // 2
	print(json.dumps(dict(value=pairs * reps / el, unit='tests/s', cores=cores, kind='port',
						  sample='{} pass(es) of coex on {} genes x {} cells fp64 ({} pairs each) in {:.1f} s; CPU oracle tile loop '
						  '(500x500 tiles, per-tile residualisation as association.py:224-249), thread pool nth={}, BLAS threads=1'.format(
							  reps, ng, n_cells, pairs, el, cores))))


def cpu_baseline(ng, n_cells, nc, seed, min_seconds=10.0):
	import subprocess
	env = dict(os.environ)
	for k in ('OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS', 'OMP_NUM_THREADS'):
		env[k] = '1'
	env['HIP_VISIBLE_DEVICES'] = ''
	r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(ng), str(n_cells), str(nc), str(seed), str(min_seconds)],
					   env=env, stdout=subprocess.PIPE, text=True, timeout=600)
	return json.loads(r.stdout.strip().splitlines()[-1])


def bench_de(args, torch, nd, world, rank, device):
	"""Extra measurements on the de shapes of BASELINE configs[2] (1 x 20k x 100k, 20 covariates: HBM-bound streaming
	path) and configs[3] (1k gRNAs x 15k genes x 50k cells: MFMA-bound general path); gene rows sharded over ranks."""
	if args.workload == 'de_c3':
		nx, ny, n, nc, seed = 1, 20000, 100000, args.covariates, 3
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

	def barrier():
		if world > 1:
			torch.distributed.barrier()
		torch.cuda.synchronize()
	for _ in range(args.warmup):
		plan.step()
	barrier()
	t0 = time.perf_counter()
	for _ in range(args.steps):
		plan.step(timed=True)
	barrier()
	elapsed = time.perf_counter() - t0
	if world > 1:
		tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
		torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
		elapsed = float(tmax.item())
	tests = nx * ny_local * world
	if rank == 0:
		ms = plan.step_ms()
		if plan.streaming():
			byts = 4.0 * n * ny_local  # algorithmic: every fp32 expression value read once
			roof = dict(bound='hbm', kernel='k_gram_skinny + sweep (whole step)', achieved=byts / (ms * 1e-3) / 1e9, peak=8000.0, unit='GB/s',
						frac=byts / (ms * 1e-3) / 8e12, traffic=None, step_ms=ms)
		else:
			fl = 2.0 * n * nx * ny_local
			roof = dict(bound='mfma', kernel='k_gram_f64 (whole step)', achieved=fl / (ms * 1e-3) / 1e12, peak=F64_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
						frac=fl / (ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, traffic=None, step_ms=ms)
		print(json.dumps(dict(metric='association tests/sec (de)', value=tests * args.steps / elapsed, unit='tests/s', n_gpus=world, steps=args.steps,
							  warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling='strong', vs_baseline=None,
							  dtype='f64', data='synthetic', config=dict(workload='norm.de {} x {} genes x {} cells, fp32 input, {} covariates'.format(nx, ny, n, nc),
							  parallelism='gene rows of Y x{}'.format(world)), roofline=roof, cpu_baseline=None)))


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument('--gpus', type=int, default=1)
	ap.add_argument('--steps', type=int, default=20)
	ap.add_argument('--warmup', type=int, default=3)
	ap.add_argument('--genes', type=int, default=5000, help='genes at N=1 (scaled by sqrt(N) for N>1)')
	ap.add_argument('--cells', type=int, default=10000)
	ap.add_argument('--cpu-seconds', type=float, default=10.0, help='minimum CPU-baseline time (0 = skip)')
	ap.add_argument('--cpu-worker', nargs=5, default=None, help=argparse.SUPPRESS)
	ap.add_argument('--workload', default='coex_c2', choices=['coex_c2', 'de_c3', 'de_c4'],
					help='coex_c2 = BASELINE configs[1] (the headline line); de_c3 / de_c4 = configs[2] / configs[3] shapes (extra measurements)')
	ap.add_argument('--covariates', type=int, default=20, help='covariates of the de_c3 workload (<= 15 selects the half-width streaming kernel)')
	ap.add_argument('--seed', type=int, default=2)
	ap.add_argument('--e2e', type=int, default=2, help='repetitions of the numpy-in/numpy-out end-to-end timing (0 = skip)')
	args = ap.parse_args()
	if args.cpu_worker:
		w = args.cpu_worker
		cpu_baseline_worker(int(w[0]), int(w[1]), int(w[2]), int(w[3]), float(w[4]))
		return

	world = int(os.environ.get('WORLD_SIZE', '1'))
	rank = int(os.environ.get('RANK', '0'))
	local_rank = int(os.environ.get('LOCAL_RANK', '0'))
	cpu = None
	if world == 1 and args.cpu_seconds > 0:
		# CPU baseline first, in a child process, before this process touches the GPU
		cpu = cpu_baseline(int(round(args.genes)), args.cells, 3, args.seed, args.cpu_seconds)

	import torch
	from normalisr_amd import distributed as nd
	if args.gpus != world:
		if world == 1 and args.gpus > 1:
			raise SystemExit('launch with: python -m torch.distributed.run --nproc-per-node {} bench.py --gpus {}'.format(args.gpus, args.gpus))
	backend = os.environ.get('NRM_DIST_BACKEND', 'nccl')  # 'gloo' + NRM_SHARE_GPU=1: functional test of the N>1 path on a 1-GPU box
	if os.environ.get('NRM_SHARE_GPU') == '1':
		local_rank = 0
	torch.cuda.set_device(local_rank)
	device = torch.device('cuda', local_rank)
	group = None
	if world > 1:
		import torch.distributed as dist
		if backend == 'nccl':
			dist.init_process_group('nccl', device_id=device)
		else:
			dist.init_process_group(backend)
		group = dist.group.WORLD

	if args.workload != 'coex_c2':
		bench_de(args, torch, nd, world, rank, device)
		if world > 1:
			torch.distributed.destroy_process_group()
		return

	n = args.cells
	# weak scaling: pairs per GPU fixed -> genes ~ sqrt(N); rounded so every rank owns the same number of rows
	ng = int(round(args.genes * np.sqrt(world) / world)) * world
	rows_local = ng // world
	dt_local, dc = synth_c2(rows_local, n, args.seed, device, torch, row0=rank * rows_local)
	e2e = None
	if world == 1 and args.e2e > 0:
		# numpy in -> numpy out through the drop-in API (H2D + kernels + D2H over PCIe); reported beside `value`, never as it
		import normalisr_amd.normalisr as norm
		# (measured before any timing event exists in this process: after hipEvents with timing have been recorded,
		#  cross-stream copies of the same process run several times slower on ROCm 7.2 -- a bench artefact, not an API cost)
		h_dt, h_dc = dt_local.cpu().numpy(), dc.cpu().numpy()
		norm.coex(h_dt[:256], h_dc)
		ts = []
		for _ in range(args.e2e):
			res = None  # the previous results are released outside the timed region
			t1 = time.perf_counter()
			res = norm.coex(h_dt, h_dc)
			ts.append(time.perf_counter() - t1)
		e2e = dict(seconds=min(ts), all_seconds=[round(t, 5) for t in ts], tests_per_s=rows_local * (rows_local - 1) // 2 / min(ts), note='norm.coex(numpy fp32) -> numpy, pageable host memory, PCIe inclusive')
		res = h_dt = None

	plan = nd.CoexPlan(dt_local, dc, rank=rank, world=world, group=group)

	def barrier():
		if world > 1:
			torch.distributed.barrier()
		torch.cuda.synchronize()

	for _ in range(args.warmup):
		plan.step()
	barrier()
	# N = 1: the kernels are bracketed by HIP events inside the timed region (one stream, the events cost nothing).
	# N > 1: the exchange runs on RCCL's stream; timing events recorded on the launch stream were measured to slow
	# cross-stream work of the same process on ROCm 7.2 (see the end_to_end_pcie note), so the timed region runs without
	# them and the per-kernel breakdown comes from three extra steps after it.
	events_inside = world == 1
	t0 = time.perf_counter()
	for _ in range(args.steps):
		plan.step(timed=events_inside)
	barrier()
	elapsed = time.perf_counter() - t0
	if not events_inside:
		for _ in range(3):
			plan.step(timed=True)
		barrier()
	if world > 1:
		tmax = torch.tensor([elapsed], device=device if backend == 'nccl' else 'cpu', dtype=torch.float64)
		torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
		elapsed = float(tmax.item())
	tests = ng * (ng - 1) // 2
	value = tests * args.steps / elapsed
	gram_ms = plan.gram_ms()  # average duration of the dominant kernel launch(es) per step on this rank
	local_pairs = plan.local_pair_count()
	traffic, traffic_src = None, None
	if world == 1 and ng == 5000 and n == 10000 and os.path.exists(PMC_FILE):
		# HBM-side bytes per k_gram_f64 launch from the committed PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)
		try:
			with open(PMC_FILE) as f:
				traffic = json.load(f)['k_gram_f64']['hbm_bytes_per_launch']
			traffic_src = 'profiles/r01_pmc_c2.json (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE, separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)'
		except (KeyError, ValueError):
			traffic = None
	if rank == 0:
		flops = 2.0 * n * local_pairs  # algorithmic: 2 n_cell flop per test, tests this rank's launches cover
		achieved = flops / (gram_ms * 1e-3) / 1e12
		out = dict(metric='association tests/sec (gene x gene coex)', value=value, unit='tests/s', n_gpus=world,
				   steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
				   scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
				   config=dict(workload='norm.coex gene x gene, {} genes x {} cells, fp32 input, 3 covariates (BASELINE configs[1]{})'.format(
					   ng, n, '' if world == 1 else ', genes scaled by sqrt(N)'), genes=ng, cells=n, covariates=3,
					   tests_per_step=tests, parallelism='gene-row blocks x{}'.format(world)),
				   roofline=dict(bound='mfma', kernel='k_gram_f64', achieved=achieved, peak=F64_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
								 frac=achieved / F64_MFMA_PEAK_TFLOPS, traffic=traffic, traffic_unit='bytes/launch', traffic_source=traffic_src,
								 algorithmic_bytes=8.0 * plan.rows_pad * plan.k_pad, kernel_ms=gram_ms),
				   kernels_ms=plan.kernel_breakdown())
		out['cpu_baseline'] = cpu
		out['end_to_end_pcie'] = e2e
		print(json.dumps(out))
	if world > 1:
		torch.distributed.destroy_process_group()


if __name__ == '__main__':
	main()
