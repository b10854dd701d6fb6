"""Multi-GPU sharding of the pair space: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI).  The reference's only parallelism is a thread-pool map over independent (x-block, y-block)
tiles (association.py:890-909,997; parallel.py:12-74); here the same independence is used across GPUs.

coex (dy=None): gene-row block b lives on rank b.  Each rank residualises its own block (covariates are
tiny and replicated), then ONE exchange step -- an all-gather of the residualised blocks and their sums
of squares -- after which rank b contracts block pairs (b, b+k mod N), k = 0..floor(N/2); for even N
the k = N/2 pair is shared half/half by its two owners.  Every unordered block pair is computed exactly
once (association.py:893-894 keeps x0 <= y0 the same way).  K (cells) is never split across GPUs, so
there is no all-reduce.  xGMI is point-to-point: an all-gather in which every GPU pushes its shard to its
7 peers at once uses all links in parallel (shard bytes / ~153 GB/s).

de (dy given): gene rows of Y are sharded, the (few) design rows X are residualised redundantly on every
rank -- no collective at all (see DePlan).
"""
import os

import numpy as np

from ._lib import ROW_TILE, K_TILE


def _round_up(v, m):
	return (v + m - 1) // m * m


def block_pair_schedule(rank, world, rows_pad):
	"""Block pairs rank `rank` contracts: list of (bi, bj, row_lo, row_hi, symmetric).
	Rows [row_lo, row_hi) of block bi (padded row indices, multiples of ROW_TILE) against all of block bj."""
	sched = [(rank, rank, 0, rows_pad, True)]
	for k in range(1, (world - 1) // 2 + 1):
		sched.append((rank, (rank + k) % world, 0, rows_pad, False))
	if world % 2 == 0 and world > 1:
		other = (rank + world // 2) % world
		lo, hi = min(rank, other), max(rank, other)
		half = _round_up(rows_pad // 2, ROW_TILE)
		if rank == lo:
			sched.append((lo, hi, 0, half, False))
		elif half < rows_pad:
			sched.append((lo, hi, half, rows_pad, False))
	return sched


def schedule_covers_all_pairs(world, rows_pad):
	"""Host-side invariant used by the tests: every unordered block pair (and every row of it) exactly once."""
	seen = {}
	for r in range(world):
		for bi, bj, lo, hi, sym in block_pair_schedule(r, world, rows_pad):
			key = (min(bi, bj), max(bi, bj))
			assert (bi <= bj) or not sym
			seen.setdefault(key, []).append((bi, bj, lo, hi))
	for a in range(world):
		for b in range(a, world):
			parts = sorted(seen.get((a, b), []), key=lambda t: t[2])
			assert parts, (a, b)
			assert len({(p[0], p[1]) for p in parts}) == 1, 'mixed orientation'
			assert parts[0][2] == 0 and parts[-1][3] == rows_pad
			for p, q in zip(parts, parts[1:]):
				assert p[3] == q[2]
	return True


class HipBackend:
	"""Block operations on the local GPU through the C ABI (normalisr_amd.engine)."""

	def __init__(self, device):
		from .engine import get_engine
		self.eng = get_engine(device)
		self.torch = self.eng.torch

	def covariates(self, dc):
		"""dc: (nc, n) device or host array -> replicated fp64 device covariates, pseudo-inverse, rank."""
		from .association import _prepare_covariates
		dc_h = dc.detach().cpu().numpy() if hasattr(dc, 'detach') else np.asarray(dc)
		dc64, dci, dcr = _prepare_covariates(dc_h)
		d_c, d_dci = self.eng.covariates(dc64, dci)
		return d_c, d_dci, dcr

	def residualize(self, x, cov, rows_pad):
		d_c, d_dci, dcr = cov
		r = self.eng.residualize(x, d_c, d_dci, dcr, rows_pad=rows_pad)
		return r.data, r.ss

	def gram(self, a, b, symmetric, rows_a=None, rows_b=None):
		from .engine import Residualized
		return self.eng.gram(Residualized(a.shape[0] if rows_a is None else rows_a, a.shape[1], a, None, None),
							 Residualized(b.shape[0] if rows_b is None else rows_b, b.shape[1], b, None, None), symmetric)

	def sweep(self, dot, ssx, ssy, nx, ny, n_cells, dof, symmetric, out_dtype, flags=None):
		p, stat, _, _, flags = self.eng.sweep(dot, ssx, ssy, nx, ny, n_cells, dof, symmetric, 0, out_dtype, flags=flags)
		return p, stat, flags

	def empty(self, shape):
		return self.torch.empty(shape, dtype=self.torch.float64, device=self.eng.device)

	def event(self):
		return self.torch.cuda.Event(enable_timing=True)

	def sync(self):
		self.torch.cuda.synchronize(self.eng.device)


class CoexPlan:
	"""Sharded coex over `world` ranks; world == 1 is the plain single-GPU path.

	dt_local: this rank's gene rows (rows_local, n) already on the device (or host for a host backend);
	every rank must own the same number of rows.  dc: (nc, n) covariates (replicated).
	step() runs one full pass and leaves the outputs of this rank's block pairs in self.outputs:
	list of dict(bi, bj, row_lo, nx, ny, symmetric, p, stat)."""

	def __init__(self, dt_local, dc, rank=0, world=1, group=None, backend=None, dimreduce=0, out_dtype=None):
		self.rank, self.world, self.group = rank, world, group
		self.be = backend if backend is not None else HipBackend(dt_local.device.index)
		self.x = dt_local
		self.rows, self.n = dt_local.shape
		self.rows_pad = _round_up(max(self.rows, 1), ROW_TILE)
		self.k_pad = _round_up(self.n, K_TILE)
		self.cov = self.be.covariates(dc)
		self.dof = self.n - 1 - self.cov[2] - dimreduce
		if self.dof <= 0:
			raise ValueError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
		if out_dtype is None:
			out_dtype = np.float32 if 'float32' in str(dt_local.dtype) else np.float64
		self.out_dtype = out_dtype
		self.sched = block_pair_schedule(rank, world, self.rows_pad)
		self.outputs = []
		self.flags = None
		self._pending = []
		self._ev = dict(residualize=[], exchange=[], gram=[], sweep=[])
		self._timed_steps = 0
		# what travels over xGMI: the fp64 residual blocks, or -- when the input is narrower than fp64 -- the raw input
		# blocks (half the bytes for fp32; partner blocks are then residualised again locally, K1 is HBM-cheap)
		self.exchange_raw = world > 1 and backend is None and 'float32' in str(dt_local.dtype)
		if world > 1:
			if self.exchange_raw:
				# blocks 0..world-1 as gathered, then copies of the first blocks so that the partners rank+1..rank+K of any
				# rank are one contiguous run of rows (one K1, one K2 and one K3 launch for all of them)
				self.n_partners = (world - 1) // 2
				self.all_x = self.be.torch.empty(((world + self.n_partners) * self.rows, self.n), dtype=dt_local.dtype, device=dt_local.device)
				self._blocks = {}
			else:
				self.all_data = self.be.empty((world * self.rows_pad, self.k_pad))
				self.all_ss = self.be.empty((world * self.rows_pad, ))

	def _timed(self, name, timed, fn):
		if not timed:
			return fn()
		e0, e1 = self.be.event(), self.be.event()
		e0.record()
		out = fn()
		e1.record()
		self._ev[name].append((e0, e1))
		return out

	def _exchange(self, data, ss):
		"""Start the all-gather of residual blocks and sums of squares; returns handles to wait on (RCCL runs it on its
		own stream, so the diagonal block pair -- local data only -- is contracted while the shards travel over xGMI)."""
		import torch.distributed as dist
		nccl = dist.get_backend(self.group) == 'nccl'
		if self.exchange_raw:
			self._blocks = {}
			if nccl:
				return [dist.all_gather_into_tensor(self.all_x[:self.world * self.rows], self.x, group=self.group, async_op=True)]
			dist.all_gather(list(self.all_x[:self.world * self.rows].view(self.world, self.rows, self.n).unbind(0)), self.x.contiguous(), group=self.group)
			return []
		if nccl:
			return [dist.all_gather_into_tensor(self.all_data, data, group=self.group, async_op=True),
					dist.all_gather_into_tensor(self.all_ss, ss, group=self.group, async_op=True)]
		dist.all_gather(list(self.all_data.view(self.world, self.rows_pad, self.k_pad).unbind(0)), data, group=self.group)
		dist.all_gather(list(self.all_ss.view(self.world, self.rows_pad).unbind(0)), ss, group=self.group)
		return []

	def block(self, b):
		if self.world == 1 or b == self.rank:
			return self._data, self._ss  # own block: local buffers (valid before the exchange has landed)
		if self.exchange_raw:
			if b not in self._blocks:  # partner block arrived raw: residualise it here (once per step)
				self._blocks[b] = self.be.residualize(self.all_x[b * self.rows:(b + 1) * self.rows], self.cov, self.rows_pad)
			return self._blocks[b]
		return (self.all_data[b * self.rows_pad:(b + 1) * self.rows_pad], self.all_ss[b * self.rows_pad:(b + 1) * self.rows_pad])

	def _pair(self, outs, timed, bi, bj, lo, hi, sym):
		a, ssa = self.block(bi)
		b, ssb = self.block(bj)
		a, ssa = a[lo:hi], ssa[lo:hi]
		nx = max(0, min(hi, self.rows) - lo)
		ny = self.rows
		if nx == 0:
			return
		dot = self._timed('gram', timed, lambda: self.be.gram(a, b, sym, nx, ny))
		p, stat, self.flags = self._timed('sweep', timed, lambda: self.be.sweep(dot, ssa, ssb, nx, ny, self.n, self.dof, sym, self.out_dtype, self.flags))
		outs.append(dict(bi=bi, bj=bj, row_lo=lo, nx=nx, ny=ny, symmetric=sym, p=p, stat=stat))

	def _partners_merged(self, outs, timed):
		"""All full block pairs (rank, rank+k), k = 1..K, as ONE rectangular problem: the partners' raw rows are a contiguous
		run of the gathered buffer, residualised by one K1 launch, contracted against the own block by one K2 launch
		(K x more tiles per launch: the persistent schedule stays in its whole-tile regime) and swept by one K3 launch."""
		R, W, K = self.rows, self.world, self.n_partners
		wrap = self.rank + K - (W - 1)
		if wrap > 0:
			self.all_x[W * R:(W + wrap) * R].copy_(self.all_x[:wrap * R])
		xs = self.all_x[(self.rank + 1) * R:(self.rank + 1 + K) * R]
		pd, pss = self._timed('residualize', timed, lambda: self.be.residualize(xs, self.cov, _round_up(K * R, ROW_TILE)))
		dot = self._timed('gram', timed, lambda: self.be.gram(self._data, pd, False, R, K * R))
		p, stat, self.flags = self._timed('sweep', timed, lambda: self.be.sweep(dot, self._ss, pss, R, K * R, self.n, self.dof, False, self.out_dtype, self.flags))
		for j in range(K):
			outs.append(dict(bi=self.rank, bj=(self.rank + 1 + j) % W, row_lo=0, nx=R, ny=R, symmetric=False,
							 p=p[:, j * R:(j + 1) * R], stat=stat[:, j * R:(j + 1) * R]))

	def step(self, timed=False):
		if timed:
			self._timed_steps += 1
		self._pending = []
		if self.world > 1 and self.exchange_raw:
			self._pending = self._exchange(None, None)  # raw rows travel: nothing to wait for, start before K1
		data, ss = self._timed('residualize', timed, lambda: self.be.residualize(self.x, self.cov, self.rows_pad))
		self._data, self._ss = data, ss
		if self.world > 1 and not self.exchange_raw:
			self._pending = self._exchange(data, ss)
		outs = []
		merged = self.exchange_raw and self.n_partners >= 1 and os.environ.get('NRM_MERGE_PARTNERS', '1') != '0'
		merged_done = False
		for bi, bj, lo, hi, sym in self.sched:
			if self._pending and not (bi == self.rank and bj == self.rank):
				def wait():
					for w in self._pending:
						w.wait()
				self._timed('exchange', timed, wait)  # time the part of the exchange that compute did not hide
				self._pending = []
			if merged and not sym and bi == self.rank and lo == 0 and hi == self.rows_pad and (bj - bi) % self.world <= self.n_partners:
				if not merged_done:
					self._partners_merged(outs, timed)
					merged_done = True
				continue
			self._pair(outs, timed, bi, bj, lo, hi, sym)
		for w in self._pending:
			w.wait()
		self._pending = []
		self.outputs = outs
		return outs

	# ---- accounting for bench.py -----------------------------------------------------------------
	def local_pair_count(self):
		"""Unique unordered gene pairs covered by this rank's Gram launches."""
		cnt = 0
		for bi, bj, lo, hi, sym in self.sched:
			nx = max(0, min(hi, self.rows) - lo)
			cnt += nx * (nx - 1) // 2 if sym else nx * self.rows
		return cnt

	def _avg_ms(self, name):
		ev = self._ev[name]
		if not ev:
			return 0.0
		self.be.sync()
		per_step = max(1, self._timed_steps)
		return sum(a.elapsed_time(b) for a, b in ev) / per_step

	def gram_ms(self):
		return self._avg_ms('gram')

	def kernel_breakdown(self):
		return {k: round(self._avg_ms(k), 4) for k in self._ev}

	# ---- assembly (validation / numpy out) -------------------------------------------------------
	def assemble(self, to_numpy):
		"""Gather every rank's output blocks on rank 0 and build the full symmetric (p, dot) matrices with
		zero diagonals.  to_numpy converts a backend array to numpy.  Returns (p, dot, var) on rank 0, None elsewhere."""
		mine = [dict(bi=o['bi'], bj=o['bj'], row_lo=o['row_lo'], symmetric=o['symmetric'], p=to_numpy(o['p']), stat=to_numpy(o['stat']))
				for o in self.outputs]
		var = to_numpy(self._ss)[:self.rows] / float(self.n)
		var[var == 0] = 1
		if self.world > 1:
			import torch.distributed as dist
			gathered = [None] * self.world if self.rank == 0 else None
			dist.gather_object((mine, var), gathered, dst=0, group=self.group)
			if self.rank != 0:
				return None
		else:
			gathered = [(mine, var)]
		ng = self.world * self.rows
		P = np.zeros((ng, ng), dtype=self.out_dtype)
		D = np.zeros((ng, ng), dtype=self.out_dtype)
		V = np.concatenate([g[1] for g in gathered]).astype(self.out_dtype)
		R = self.rows
		for blocks, _ in gathered:
			for o in blocks:
				r0 = o['bi'] * R + o['row_lo']
				c0 = o['bj'] * R
				nx, ny = o['p'].shape
				P[r0:r0 + nx, c0:c0 + ny] = o['p']
				D[r0:r0 + nx, c0:c0 + ny] = o['stat']
				if not o['symmetric']:
					P[c0:c0 + ny, r0:r0 + nx] = o['p'].T
					D[c0:c0 + ny, r0:r0 + nx] = o['stat'].T
		return P, D, V


class DePlan:
	"""Sharded de (dy given): rank r owns a block of gene rows of Y; the design rows X and the covariates are
	replicated and residualised redundantly, so there is NO collective on the data path (outputs are disjoint
	column blocks of the (n_x, n_y) result).  step() runs one resident pass (outputs stay in HBM, self.result);
	results() returns this rank's (p, gamma, varx, vary) as numpy arrays."""

	def __init__(self, dx, dy_local, dc, rank=0, world=1, dimreduce=0, return_dot=False, device=None):
		from .association import _prepare_covariates
		from .engine import get_engine
		self.rank, self.world = rank, world
		self.eng = get_engine(device)
		self.dx, self.dy = dx, dy_local
		dc_h = dc.detach().cpu().numpy() if hasattr(dc, 'detach') else np.asarray(dc)
		self.dc64, self.dci, self.dcr = _prepare_covariates(dc_h)
		self.dimreduce, self.return_dot = dimreduce, return_dot
		self.out_dtype = np.float32 if 'float32' in str(dy_local.dtype) else np.float64
		self.nx, self.n = dx.shape
		self.ny = dy_local.shape[0]
		if self.n <= self.dcr + dimreduce + 1:
			raise ValueError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
		self.cov = self.eng.covariates(self.dc64, self.dci)  # resident on the device across steps
		self.result = None
		self._ev = []

	def step(self, timed=False):
		torch = self.eng.torch
		if timed:
			e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			e0.record()
		# resident step: p / gamma / sums of squares stay in HBM; results() brings them to the host and checks the flags
		self.result = self.eng.association_single0(self.dx, self.dy, self.dc64, self.dci, self.dcr, self.dimreduce,
												   return_dot=self.return_dot, want_alpha=False, out_dtype=self.out_dtype, cov=self.cov,
												   resident=True)
		if timed:
			e1.record()
			self._ev.append((e0, e1))
		return self.result

	def results(self):
		"""(p, gamma, varx, vary) of the last step as numpy arrays, after the reference's assertions (association.py:248,252)."""
		r = self.result
		self.eng.check_flags(r['flags'])
		return (self.eng.download(r['p']), self.eng.download(r['stat']), self.eng.variances(r['ssx'], self.nx, self.n, self.out_dtype),
				self.eng.variances(r['ssy'], self.ny, self.n, self.out_dtype))

	def step_ms(self):
		self.eng.torch.cuda.synchronize()
		return sum(a.elapsed_time(b) for a, b in self._ev) / max(1, len(self._ev))

	def streaming(self):
		return self.eng.de_streaming_ok(self.dx, self.dy, self.dc64)


def coex(dt_local, dc, group=None, dimreduce=0):
	"""Sharded norm.coex for one-process-per-GPU programs: every rank passes ITS block of gene rows (same row count on
	every rank; numpy or a torch tensor on its GPU) and the replicated covariates.  Rank 0 gets (p, dot, var) as numpy
	arrays with the reference's contract (symmetric, zero diagonals; coex.py:4-48); other ranks get None.

	    # torchrun --nproc-per-node 8 script.py
	    dist.init_process_group('nccl'); torch.cuda.set_device(local_rank)
	    res = normalisr_amd.distributed.coex(dt[rank * R:(rank + 1) * R], dc)
	"""
	import torch
	import torch.distributed as dist
	world = dist.get_world_size(group) if dist.is_initialized() else 1
	rank = dist.get_rank(group) if dist.is_initialized() else 0
	dev = torch.device('cuda', torch.cuda.current_device())
	if isinstance(dt_local, np.ndarray):
		a = dt_local if dt_local.dtype in (np.float32, np.float64) else dt_local.astype(np.float64)
		dt_local = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
	plan = CoexPlan(dt_local, dc, rank=rank, world=world, group=group if group is not None else (dist.group.WORLD if world > 1 else None),
					dimreduce=dimreduce)
	plan.step()
	plan.be.eng.check_flags(plan.flags)
	return plan.assemble(lambda t: t.detach().cpu().numpy())


def de(dg, dt_local, dc, group=None, dimreduce=0):
	"""Sharded norm.de (single=0) for one-process-per-GPU programs: every rank passes the full grouping matrix dg, ITS
	block of gene rows of dt (numpy or a torch tensor on its GPU) and the replicated covariates.  No collective on the data
	path; rank 0 gathers the column blocks and returns (p, gamma, None, varg, vart) with the reference's contract
	(de.py:4-132, constant groupings re-inflated), other ranks get None."""
	import torch
	import torch.distributed as dist
	from .de import _varying_rows
	world = dist.get_world_size(group) if dist.is_initialized() else 1
	rank = dist.get_rank(group) if dist.is_initialized() else 0
	dev = torch.device('cuda', torch.cuda.current_device())
	dg = np.asarray(dg)
	gid = _varying_rows(dg)  # de.py:93
	if isinstance(dt_local, np.ndarray):
		a = dt_local if dt_local.dtype in (np.float32, np.float64) else dt_local.astype(np.float64)
		dt_local = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
	x = dg[gid]
	x = x if x.dtype in (np.float32, np.float64) else x.astype(np.float64)
	plan = DePlan(torch.from_numpy(np.ascontiguousarray(x)).to(dev), dt_local, dc, rank=rank, world=world, dimreduce=dimreduce)
	plan.step()
	mine = plan.results()
	if world > 1:
		gathered = [None] * world if rank == 0 else None
		dist.gather_object(mine, gathered, dst=0, group=group)
		if rank != 0:
			return None
	else:
		gathered = [mine]
	odt = plan.out_dtype
	p = np.concatenate([g[0] for g in gathered], axis=1)
	gam = np.concatenate([g[1] for g in gathered], axis=1)
	vt = np.concatenate([g[3] for g in gathered])
	ng0, nt = dg.shape[0], p.shape[1]
	P = np.ones((ng0, nt), dtype=odt)
	G = np.zeros((ng0, nt), dtype=odt)
	VG = np.zeros((ng0, ), dtype=odt)
	VT = np.zeros((ng0, nt), dtype=odt)
	P[gid], G[gid], VG[gid], VT[gid] = p, gam, gathered[0][2], vt  # de.py:107-122
	return (P, G, None, VG, VT)
