"""CPU tests of the N>1 host logic: the block-pair schedule covers every unordered pair exactly once, and
a world_size-2 gloo run of CoexPlan (exchange + schedule + assembly) reproduces the single-process
oracle.  The block arithmetic is supplied by a numpy backend built on the oracle (test infrastructure);
the HIP backend is exercised by the gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle
from conftest import ROOT, relerr
from normalisr_amd.distributed import CoexPlan, TensorBlocks, block_pair_schedule, schedule_covers_all_pairs


@pytest.mark.parametrize('world', [1, 2, 3, 4, 5, 8])
def test_schedule_covers_every_pair_once(world):
	for rows_pad in (128, 384, 640):
		assert schedule_covers_all_pairs(world, rows_pad)
	# balance: block-pair work differs by at most one half block between ranks
	work = []
	for r in range(world):
		work.append(sum((hi - lo) * (0.5 if sym else 1.0) for _, _, lo, hi, sym in block_pair_schedule(r, world, 640)))
	assert max(work) - min(work) <= 128 + 1e-9


class OracleBackend(TensorBlocks):
	"""Block operations in numpy (oracle arithmetic) on CPU torch tensors, for gloo tests only."""

	def __init__(self):
		import torch
		self.torch = torch

	def covariates(self, dc):
		dc = np.asarray(dc, dtype=np.float64)
		if dc.shape[0] and (dc != 0).any():
			dci, dcr = oracle.inv_rank(dc @ dc.T)
		else:
			dci, dcr = np.zeros((dc.shape[0], ) * 2), 0
		return dc, dci, dcr

	def residualize(self, x, cov, rows_pad, chunks=0):
		self._nchunks = max(1, chunks)
		dc, dci, dcr = cov
		x = np.asarray(x, dtype=np.float64)
		r = x - (dci @ (dc @ x.T)).T @ dc if dcr > 0 else x
		kp = (x.shape[1] + 15) // 16 * 16
		out = np.zeros((rows_pad, kp))
		out[:x.shape[0], :x.shape[1]] = r
		return self.torch.from_numpy(out), self.torch.from_numpy((out**2).sum(axis=1))

	def gram(self, a, b, symmetric, rows_a=None, rows_b=None):
		return self.torch.from_numpy(a.numpy() @ b.numpy().T)

	def sweep(self, dot, ssx, ssy, nx, ny, n_cells, dof, symmetric, out_dtype, flags=None, a=None, b=None):
		d = dot.numpy()[:nx, :ny]
		sx, sy = ssx.numpy()[:nx].copy(), ssy.numpy()[:ny].copy()
		sx[sx == 0] = n_cells
		sy[sy == 0] = n_cells
		p = oracle.pvalues(d * d / np.outer(sx, sy), dof)
		stat = d / n_cells
		if symmetric:
			p = np.triu(p, 1) + np.triu(p, 1).T
			stat = np.triu(stat, 1) + np.triu(stat, 1).T
		return p.astype(out_dtype), stat.astype(out_dtype), flags

	def event(self):
		raise NotImplementedError

	def sync(self):
		pass


def _worker(rank, world, port, q):
	import torch.distributed as dist
	sys.path.insert(0, ROOT)
	sys.path.insert(0, os.path.join(ROOT, 'tests'))
	dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{}'.format(port), rank=rank, world_size=world)
	rng = np.random.default_rng(42)
	ng, n = 200 * world, 150
	dt = rng.normal(size=(ng, n)) + 0.5 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n))
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	R = ng // world
	plan = CoexPlan(dt[rank * R:(rank + 1) * R], dc, rank=rank, world=world, group=dist.group.WORLD, backend=OracleBackend(),
					out_dtype=np.float64)
	assert plan.chunks == (4 if os.environ['NRM_EXCHANGE'] == 'chunks' else 0)
	plan.step()
	res = plan.assemble()
	if rank == 0:
		q.put(tuple(np.array(a) for a in res))
	dist.barrier()
	dist.destroy_process_group()


def _free_port():
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	p = s.getsockname()[1]
	s.close()
	return p


@pytest.mark.parametrize('world,exchange', [(2, 'chunks'), (3, 'chunks'), (4, 'chunks'), (2, 'blocks'), (3, 'blocks')])
def test_gloo_sharded_coex_matches_single_process(world, exchange, monkeypatch):
	"""Exchange, schedule, completion of every rank's row block by point-to-point messages (mirrored blocks, the half-split
	pair of an even world) and assembly through arrays shared by the ranks -- nothing is pickled through rank 0.
	exchange = chunks: the pipelined form (4 cell chunks gathered one after another, block pairs accumulated chunk by chunk);
	blocks: one all-gather of whole residualised blocks."""
	import torch.multiprocessing as mp
	monkeypatch.setenv('NRM_EXCHANGE', exchange)  # inherited by the spawned ranks
	monkeypatch.setenv('NRM_EXCHANGE_MIN_KSTEPS', '1')
	monkeypatch.setenv('NRM_EXCHANGE_CHUNKS', '4')
	ctx = mp.get_context('spawn')
	q = ctx.Queue()
	port = _free_port()
	procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
	for p in procs:
		p.start()
	from conftest import queue_get
	P, D, V = queue_get(q, procs)
	for p in procs:
		p.join(timeout=120)
		assert p.exitcode == 0
	rng = np.random.default_rng(42)
	ng, n = 200 * world, 150
	dt = rng.normal(size=(ng, n)) + 0.5 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n))
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	po, do, vo = oracle.coex(dt, dc)
	assert relerr(P, po, 1e-300) < 1e-9 and relerr(D, do, 1e-13) < 1e-9 and relerr(V, vo) < 1e-12
	assert (np.diag(P) == 0).all() and (P == P.T).all() and (D == D.T).all()
