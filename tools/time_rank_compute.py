"""Compute side of ONE rank's step of the sharded coex bench (weak scaling of configs[1]: genes x sqrt(N)) on a single GPU:
the rank's K1, every Gram launch of its block-pair schedule (chunked, partners merged as CoexPlan does) and its sweeps, with the
gather buffers filled locally instead of over xGMI.  What N > 1 costs in launch granularity, before any communication.
Usage: time_rank_compute.py [genes cells]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd.distributed import HipBackend, block_pair_schedule, _round_up
from normalisr_amd._lib import ROW_TILE
genes, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 10000)
be = HipBackend(0)
eng = be.eng
g = torch.Generator(device='cuda').manual_seed(2)
dc = torch.cat([torch.randn((2, n), generator=g, device='cuda'), torch.ones((1, n), device='cuda')])
cov = be.covariates(dc)
base = None
for world in (1, 2, 4, 8):
	rows = int(round(genes * np.sqrt(world) / world))
	rp = _round_up(rows, ROW_TILE)
	x = torch.randn((rows, n), dtype=torch.float32, device='cuda', generator=g)
	nks = (_round_up(n, 16) + 31) // 32
	S = max(1, min(8, nks // 128)) if world > 1 else 1
	rank = 0
	sched = block_pair_schedule(rank, world, rp)
	dof = n - 1 - cov[2]

	blk0, _ = be.residualize(x, cov, rp, chunks=S)
	chunks, once = be.chunk_payload(blk0)
	g_chunks = [t.unsqueeze(0).expand(world, -1).contiguous() for t in chunks] if world > 1 else None  # what the all-gathers deliver
	g_exps = once[0].unsqueeze(0).expand(world, -1).contiguous() if world > 1 else None

	def step():
		blk, ss = be.residualize(x, cov, rp, chunks=S)
		nch = be.n_chunks(blk)
		flags = None
		full = [e for e in sched if e[0] == rank and e[1] != rank and e[2] == 0 and e[3] == rp]
		K = len(full)
		merged = K >= 2
		for bi, bj, lo, hi, sym in sched:
			if merged and (bi, bj, lo, hi, sym) in full:
				continue
			nx = max(0, min(hi, rows) - lo)
			if nx == 0:
				continue
			a = blk if (lo == 0 and hi == rp) else be.rows(blk, lo, hi, nx)
			dot = None
			for c in range(nch):
				dot = be.gram_chunk(a, blk, sym, c, dot, c > 0)
			_, _, flags = be.sweep(dot, ss[lo:hi], ss, nx, rows, n, dof, sym, np.float32, flags)
		if merged:
			mdot = None
			for c in range(nch):
				mdot = be.gram_chunk_blocks(blk, g_chunks[c], [g_exps], 1, K, c, mdot, c > 0)
			for j in range(K):
				_, _, flags = be.sweep(mdot[:, j * rp:(j + 1) * rp], ss, ss, rows, rows, n, dof, False, np.float32, flags)
	for _ in range(3):
		step()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(10):
		step()
	e1.record()
	torch.cuda.synchronize()
	ms = e0.elapsed_time(e1) / 10
	base = base or ms
	pairs = sum((max(0, min(hi, rows) - lo)) * ((max(0, min(hi, rows) - lo) - 1) / 2 if sym else rows) for bi, bj, lo, hi, sym in sched)
	print('N=%d: %d rows per rank, %d chunks, %d launches of the schedule: %.3f ms per step for %.3g pairs (%.2f of the N=1 rate)' % (
		world, rows, S, len(sched), ms, pairs, (pairs / ms) / (genes * (genes - 1) / 2 / base)))
