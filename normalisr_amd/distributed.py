"""Multi-GPU sharding of the pair space: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI).  The reference's only parallelism is a thread-pool map over independent (x-block, y-block)
tiles (association.py:890-909,997; parallel.py:12-74); here the same independence is used across GPUs.

coex (dy=None): gene-row block b lives on rank b.  Each rank residualises its own block (covariates are
tiny and replicated) and all-gathers it -- as fixed-point digit planes in cell chunks, contracted chunk by
chunk as they land (CoexPlan._step_chunked), or in one piece -- and rank b contracts block pairs
(b, b+k mod N), k = 0..floor(N/2); for even N the k = N/2 pair is shared half/half by its two owners.  Every unordered block pair is computed exactly
once (association.py:893-894 keeps x0 <= y0 the same way).  K (cells) is never split across GPUs, so
there is no all-reduce.  xGMI is point-to-point: an all-gather in which every GPU pushes its shard to its
7 peers at once uses all links in parallel (shard bytes / ~153 GB/s).

de (dy given): gene rows of Y are sharded, the (few) design rows X are residualised redundantly on every
rank -- no collective at all (see DePlan).
"""
import os

import numpy as np

from . import _opts

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


def _coll_tensor(values, group, device, dtype=None):
	"""Small tensor for a collective: on the GPU for RCCL, on the host for gloo."""
	import torch
	import torch.distributed as dist
	dev = device if dist.get_backend(group) == 'nccl' else 'cpu'
	return torch.tensor(values, dtype=torch.int64 if dtype is None else dtype, device=dev)


def _all_gather_ints(values, group, device):
	"""(world, len(values)) int64 numpy array with every rank's values."""
	import torch
	import torch.distributed as dist
	mine = _coll_tensor(list(values), group, device)
	out = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
	dist.all_gather(out, mine, group=group)
	return np.stack([t.cpu().numpy() for t in out])


def _reduce_flags(flags, group):
	"""A call's device counters (engine.new_flags) as ONE verdict of all ranks: the counts summed, the guard's largest error
	estimate (float bits in the last entry) maximised.  Returns the reduced tensor (on the host for gloo)."""
	import torch.distributed as dist
	f = flags.clone() if dist.get_backend(group) == 'nccl' else flags.cpu()
	worst = f[3:4].clone() if f.shape[0] >= 4 else None
	f[:3].clamp_(max=1 << 24)  # counters say "how many" only as a diagnostic: bounded per rank, their int32 sum cannot wrap
	dist.all_reduce(f, group=group)
	if worst is not None:
		dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=group)
		f[3:4] = worst
	return f


def _on_engine(method):
	"""Plan method that holds its engine's lock (engine.serialised): a resident plan's step is a sequence of launches on shared
	scratch buffers, so two threads stepping plans on one device take turns.  Backends without an engine (the numpy backend of the
	CPU tests) need no lock."""
	import functools

	@functools.wraps(method)
	def locked(self, *a, **ka):
		eng = getattr(self, 'eng', None) or getattr(getattr(self, 'be', None), 'eng', None)
		if eng is None:
			return method(self, *a, **ka)
		with eng.lock:
			return method(self, *a, **ka)

	return locked


class SharedArrays:
	"""Result arrays visible to every rank of the node, so that each rank copies ITS finished rows straight into the
	caller-visible output (the reference's gather loop fills one array from every tile, association.py:1005-1034) and
	nothing is pickled through rank 0.  The arrays are .npy files mapped by every rank (numpy.lib.format.open_memmap)
	in a directory created by rank 0 -- on tmpfs (/dev/shm) by default, i.e. shared memory; with keep=True in a
	directory of the caller's choice, i.e. the per-run result files.  Only the directory name travels between ranks."""

	def __init__(self, specs, rank, world, group, base=None, keep=False):
		import shutil
		import tempfile
		self.rank, self.world, self.group, self.keep = rank, world, group, keep
		path = [None, None]  # (directory, error): every rank learns of a failure on rank 0 instead of waiting in the collective
		if rank == 0:
			try:
				need = sum(int(np.prod(shape)) * np.dtype(dtype).itemsize + 4096 for shape, dtype in specs.values())
				if keep:
					os.makedirs(base, exist_ok=True)
					path[0] = base
				else:
					# tmpfs reserves nothing at creation: a /dev/shm smaller than the results (64 MB by default in a container) would
					# kill the ranks with SIGBUS on their first write -- fall back to the default temporary directory
					base = base or ('/dev/shm' if os.path.isdir('/dev/shm') and shutil.disk_usage('/dev/shm').free > need + (64 << 20) else None)
					path[0] = tempfile.mkdtemp(prefix='nrm_shared_', dir=base)
				free = shutil.disk_usage(path[0]).free
				if free < need:
					raise OSError('result arrays need {} bytes, {} has {} free'.format(need, path[0], free))
				for name, (shape, dtype) in specs.items():
					m = np.lib.format.open_memmap(os.path.join(path[0], name + '.npy'), mode='w+', dtype=np.dtype(dtype), shape=tuple(shape))
					del m
			except Exception as e:
				path = [None, '{}: {}'.format(type(e).__name__, e)]
		if world > 1:
			import torch.distributed as dist
			dist.broadcast_object_list(path, src=0, group=group)  # a directory name, not data
		if path[1] is not None:
			raise RuntimeError('SharedArrays: rank 0 could not create the result arrays ({})'.format(path[1]))
		self.dir = path[0]
		self.arrays = {name: np.lib.format.open_memmap(os.path.join(self.dir, name + '.npy'), mode='r+') for name in specs}

	def __getitem__(self, name):
		return self.arrays[name]

	def finish(self):
		"""All ranks have written: rank 0 keeps its mappings (returned to the caller) and removes the files unless keep."""
		for a in self.arrays.values():
			a.flush()
		if self.world > 1:
			import torch.distributed as dist
			dist.barrier(group=self.group)
		if self.rank != 0:
			self.arrays = {}
			return None
		if not self.keep:
			import shutil
			shutil.rmtree(self.dir, ignore_errors=True)  # the mappings stay valid until the arrays are released
		return self.arrays


class StepGraph:
	"""A resident step (same device buffers in, same kernels, results left in HBM) as a HIP graph: the first call runs eagerly, the
	second is captured (stream capture through torch: the C ABI launches on torch's current stream, torch's allocator serves the
	step's temporaries from the graph's private pool), later calls replay it -- one submission instead of ten launches, which
	matters for the 2 ms steps (de on configs[2]: the launch gaps between its small kernels were ~10 % of the step).
	NRM_GRAPH=0, or any failure to capture, leaves the step eager."""

	def __init__(self, torch):
		self.torch = torch
		self.calls = 0
		self.graph = None
		self.result = None
		self.enabled = _opts.debug('graph', '1') != '0'

	def run(self, fn):
		self.calls += 1
		if self.graph is not None:
			self.graph.replay()
			return self.result
		if not self.enabled or self.calls < 2:
			return fn()
		torch = self.torch
		try:
			torch.cuda.synchronize()
			g = torch.cuda.CUDAGraph()
			with torch.cuda.graph(g):
				res = fn()
			self.graph, self.result = g, res
			g.replay()
			return res
		except Exception as e:  # not capturable here (e.g. a host synchronisation inside the step): stay eager
			import logging
			logging.info('normalisr_amd: step not captured as a HIP graph (%s); running it eagerly.', e)
			self.enabled = False
			torch.cuda.synchronize()
			return fn()


class HipBackend:
	"""Block operations on the local GPU through the C ABI (normalisr_amd.engine).  A block is a Residualized: fp64 residual rows
	(fp64 Gram kernel) or the fixed-point digit planes K1 wrote for the integer engine, plus its sums of squares."""

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

	def chunkable(self, x, cov):
		"""The digit planes of these rows can be written (and sent) in cell chunks: integer engine and 16-byte aligned rows."""
		return bool(self.eng.gram_slices(x.shape[1])) and self.eng.k1_quantises(x, cov[0])

	def residualize(self, x, cov, rows_pad, chunks=0, into=None):
		"""into: the block a previous step of the same plan got from here, to be overwritten (engine.residualize)."""
		d_c, d_dci, dcr = cov
		ns = self.eng.gram_slices(x.shape[1])
		if chunks:
			r = self.eng.residualize_chunked(x, d_c, d_dci, dcr, rows_pad, ns, chunks, into=into)
			return r, r.ss
		r = self.eng.residualize(x, d_c, d_dci, dcr, rows_pad=rows_pad, nslices=ns, keep_fp64=not ns, into=into)
		if ns and getattr(r, '_quant', None) is None:  # rows K1 could not quantise itself (unaligned): separate pass
			self.eng.quantized(r, ns)
		return r, r.ss

	def rows(self, blk, lo, hi, rows=None):
		return self.eng.row_block(blk, lo, hi, rows)

	def payload(self, blk):
		"""What travels to the other ranks for this block: its digit planes and row exponents (6 bytes per value at 6 slices
		against 8 for fp64 residuals), or the fp64 residuals for the fp64 engine."""
		q = getattr(blk, '_quant', None)
		return [q[0], q[1], blk.fix] if q is not None else [blk.data]

	def from_payload(self, parts, rows, rows_pad, n, k_pad, ss):
		from .engine import Residualized
		if len(parts) == 3:  # digit planes, row exponents, row records (csrc/nrm_fix.h)
			blk = Residualized(rows, n, None, ss, None, shape=(rows_pad, k_pad))
			blk._quant = (parts[0], parts[1], self.eng.gram_slices(n))
			blk.fix = parts[2]
			return blk
		return Residualized(rows, n, parts[0], ss, None, shape=(rows_pad, k_pad))

	def gram(self, a, b, symmetric, rows_a=None, rows_b=None):
		return self.eng.gram(a, b, symmetric, nslices=self.eng.gram_slices(a.n))

	# cell-chunked blocks (the pipelined exchange): every chunk is an operand of its own, the row exponents are shared
	def n_chunks(self, blk):
		return len(blk._quant[0])

	def chunk_payload(self, blk):
		"""(tensors of the chunks, in cell order; tensors that travel once: the row exponents and the row records)."""
		return list(blk._quant[0]), [blk._quant[1], blk.fix]

	def from_chunks(self, chunks, once, like, ss):
		from .engine import Residualized
		blk = Residualized(like.rows, like.n, None, ss, None, shape=(like.rows_pad, like.k_pad))
		blk._quant = (list(chunks), once[0], like._quant[2])
		blk.cks = like.cks
		blk.fix = once[1]
		return blk

	def gram_chunk(self, a, b, symmetric, chunk, dot, accumulate):
		return self.eng.gram_chunk(a, b, symmetric, chunk, dot, accumulate)

	def gram_chunk_blocks(self, a, g_chunk, g_once, first, count, chunk, dot, accumulate):
		"""All full partner blocks first .. first + count - 1 (cyclic) of the gather buffers against `a` in one launch."""
		return self.eng.gram_chunk_blocks(a, g_chunk, g_once[0], first, count, chunk, dot, accumulate)

	def sweep(self, dot, ssx, ssy, nx, ny, n_cells, dof, symmetric, out_dtype, flags=None, a=None, b=None):
		"""a, b: the blocks dot was made from (their row records feed K3's correction and guard when the integer engine made it)."""
		fix = self.eng.fix_args(a, b) if a is not None and b is not None else None
		p, stat, _, _, flags = self.eng.sweep(dot, ssx, ssy, nx, ny, n_cells, dof, symmetric, 0, out_dtype, flags=flags, fix=fix)
		return p, stat, flags

	def event(self):
		return self.torch.cuda.Event(enable_timing=True)

	def sync(self):
		self.torch.cuda.synchronize(self.eng.device)


class TensorBlocks:
	"""Mixin for backends whose blocks are plain (rows_pad, k_pad) tensors (the numpy backend of the CPU tests)."""

	def rows(self, blk, lo, hi, rows=None):
		return blk[lo:hi]

	def payload(self, blk):
		return [blk]

	def from_payload(self, parts, rows, rows_pad, n, k_pad, ss):
		return parts[0]

	# cell chunks of a (rows_pad, k_pad) tensor: column ranges of equal width
	def chunkable(self, x, cov):
		return True

	def _cell_ranges(self, blk):
		w = -(-blk.shape[1] // self._nchunks)
		return [(c, min(blk.shape[1], c + w)) for c in range(0, blk.shape[1], w)]

	def n_chunks(self, blk):
		return len(self._cell_ranges(blk))

	def chunk_payload(self, blk):
		return [blk[:, a:b].contiguous() for a, b in self._cell_ranges(blk)], []

	def from_chunks(self, chunks, once, like, ss):
		import torch
		return torch.cat(list(chunks), dim=1)

	def gram_chunk(self, a, b, symmetric, chunk, dot, accumulate):
		c0, c1 = self._cell_ranges(a)[chunk]
		part = a[:, c0:c1] @ b[:, c0:c1].T
		return dot + part if accumulate else part


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
		# What travels over xGMI (NRM_EXCHANGE = auto | chunks | blocks | raw):
		#  chunks (auto, whenever the rows go to the integer engine): this rank's fixed-point digit planes, cut along the cells
		#    into NRM_EXCHANGE_CHUNKS pieces that are all-gathered one after another; the block pairs are contracted chunk by
		#    chunk as the pieces land (exact partial sums, added in fp64), so only the first piece's flight is exposed and the
		#    partner blocks are never residualised again;
		#  raw (auto for fp32 input on 2-3 ranks, or when the rows do not go to the integer engine): the raw rows, 4 bytes per value,
		#    residualised again by the receiver;
		#  blocks (auto otherwise): the residualised blocks in one piece, as the backend packs them.
		mode = os.environ.get('NRM_EXCHANGE', 'auto')
		if mode not in ('auto', 'chunks', 'blocks', 'raw'):
			raise ValueError('NRM_EXCHANGE must be auto, chunks, blocks or raw')
		# NRM_FORCE_EXCHANGE = chunks | blocks | raw: take the N > 1 code path of that exchange form on ONE rank -- the collectives run
		# (a group of one: RCCL on the one GPU of a development box), and the rank's own block pair is contracted from the GATHER
		# BUFFERS after the waits instead of from the local block, so what the collective delivered is what is computed on
		forced = _opts.debug('force_exchange', '')
		if forced and forced not in ('chunks', 'blocks', 'raw'):
			raise ValueError('NRM_FORCE_EXCHANGE must be chunks, blocks or raw')
		self.forced = bool(forced) and world == 1 and group is not None and backend is None
		if self.forced:
			mode = forced
		multi = world > 1 or self.forced
		self.multi = multi
		self.chunks = 0
		# auto: with 2 or 3 ranks a GPU talks over one or two of its seven xGMI links and the step is bound by the bytes on the
		# wire, not by how well they hide -- fp32 rows then travel raw (4 bytes per value against 6 as digit planes)
		few_links = mode == 'auto' and world <= 3 and backend is None and 'float32' in str(dt_local.dtype)
		if multi and mode in ('auto', 'chunks') and not few_links and hasattr(self.be, 'gram_chunk') and self.be.chunkable(dt_local, self.cov):
			# chunk launches of fewer than ~128 k-steps (4096 cells) cost more than they hide (tools/time_chunks.py: a 1792 x 1792
			# block pair over 10 000 cells takes 1.04x in 2 chunks, 1.33x in 4, 2.0x in 8; at 100 000 cells 4 chunks are free)
			min_ks = int(_opts.debug('exchange_min_ksteps', '128'))
			self.chunks = max(1, min(int(_opts.debug('exchange_chunks', '8')), ((self.k_pad + 31) // 32) // max(1, min_ks)))
		elif multi and mode == 'chunks':
			raise ValueError('NRM_EXCHANGE=chunks needs the integer Gram engine and 16-byte aligned rows')
		self.exchange_raw = multi and backend is None and not self.chunks and mode != 'blocks' and 'float32' in str(dt_local.dtype)
		if multi and backend is None:
			# every rank must own the same number of rows of the same dtype: a mismatch would hang the all-gather or
			# mis-slice the gathered rows, and ranks with different dtypes (or exchange modes) would issue different collectives
			code = {'torch.float32': 0, 'torch.float64': 1}.get(str(dt_local.dtype), 2)
			shapes = _all_gather_ints([self.rows, self.n, code, self.chunks, int(self.exchange_raw)], group, dt_local.device)
			if not (shapes[:, :3] == shapes[0, :3]).all():
				raise ValueError('Sharded coex needs the same (rows, cells, dtype) on every rank; got {}'.format(shapes[:, :3].tolist()))
			if not (shapes[:, 3:] == shapes[0, 3:]).all():
				# the ranks' local views differ in alignment (one can write cell chunks, another cannot): agree on the one form every
				# rank can produce -- whole blocks -- instead of issuing different collectives
				if mode == 'chunks':
					raise ValueError('NRM_EXCHANGE=chunks needs 16-byte aligned rows on every rank')
				self.chunks, self.exchange_raw = 0, False
		self._gathered = None
		if multi and self.exchange_raw:
			# blocks 0..world-1 as gathered, then copies of the first blocks so that the partners rank+1..rank+K of any
			# rank are one contiguous run of rows (one K1, one K2 and one K3 launch for all of them)
			self.n_partners = (world - 1) // 2
			self.all_x = self.be.torch.empty(((world + self.n_partners) * self.rows, self.n), dtype=dt_local.dtype, device=dt_local.device)

	def _into(self):
		"""The previous step's own block, for K1 to overwrite (nothing reads it once the step's outputs exist) -- device backend only."""
		prev = getattr(self, '_blk', None)
		return dict(into=prev) if prev is not None and isinstance(self.be, HipBackend) else {}

	def _timed(self, name, timed, fn):
		if not timed:
			return fn()
		e0, e1 = self.be.event(), self.be.event()
		e0.record()
		out = fn()
		e1.record()
		self._ev[name].append((e0, e1))
		return out

	def _exchange(self, blk, ss):
		"""Start the all-gather of this rank's block; returns handles to wait on (RCCL runs it on its own stream, so the
		diagonal block pair -- local data only -- is contracted while the shards travel over xGMI)."""
		import torch.distributed as dist
		torch = self.be.torch
		nccl = dist.get_backend(self.group) == 'nccl'
		self._partner_prev, self._partner = getattr(self, '_partner', {}), {}
		if self.exchange_raw:
			if nccl:
				return [dist.all_gather_into_tensor(self.all_x[:self.world * self.rows], self.x, group=self.group, async_op=True)]
			dist.all_gather(list(self.all_x[:self.world * self.rows].view(self.world, self.rows, self.n).unbind(0)), self.x.contiguous(), group=self.group)
			return []
		parts = [t.contiguous() for t in self.be.payload(blk)] + [ss.contiguous()]
		if self._gathered is None or any(g.shape[1:] != t.shape or g.dtype != t.dtype for g, t in zip(self._gathered, parts)):
			self._gathered = [torch.empty((self.world, ) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in parts]
		if nccl:
			return [dist.all_gather_into_tensor(g, t, group=self.group, async_op=True) for g, t in zip(self._gathered, parts)]
		for g, t in zip(self._gathered, parts):
			dist.all_gather(list(g.unbind(0)), t, group=self.group)
		return []

	def block(self, b):
		if not self.multi or (b == self.rank and not self.forced):
			return self._blk, self._ss  # own block: local buffers (valid before the exchange has landed)
		if b not in self._partner:
			if self.chunks:  # views of the gather buffers: the chunks land one after another
				ss = self._g_once[-1][b]
				self._partner[b] = (self.be.from_chunks([g[b] for g in self._g_chunks], [g[b] for g in self._g_once[:-1]], self._blk, ss), ss)
			elif self.exchange_raw:  # partner block arrived raw: residualise it here (once per step)
				prev = getattr(self, '_partner_prev', {}).get(b)
				self._partner[b] = self.be.residualize(self.all_x[b * self.rows:(b + 1) * self.rows], self.cov, self.rows_pad,
													   **(dict(into=prev[0]) if prev is not None and isinstance(self.be, HipBackend) else {}))
			else:
				ss = self._gathered[-1][b]
				self._partner[b] = (self.be.from_payload([g[b] for g in self._gathered[:-1]], self.rows, self.rows_pad, self.n, self.k_pad, ss), ss)
		return self._partner[b]

	def _pair(self, outs, timed, bi, bj, lo, hi, sym):
		a, ssa = self.block(bi)
		b, ssb = self.block(bj)
		nx = max(0, min(hi, self.rows) - lo)
		ny = self.rows
		if nx == 0:
			return
		if lo != 0 or hi != self.rows_pad:
			a, ssa = self.be.rows(a, lo, hi, nx), ssa[lo:hi]
		dot = self._timed('gram', timed, lambda: self.be.gram(a, b, sym, nx, ny))
		p, stat, self.flags = self._timed('sweep', timed, lambda: self.be.sweep(dot, ssa, ssb, nx, ny, self.n, self.dof, sym, self.out_dtype, self.flags, a=a, b=b))
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
		prev = getattr(self, '_pd', None)  # (the previous step's partner block: overwritten, see _into)
		pd, pss = self._timed('residualize', timed, lambda: self.be.residualize(xs, self.cov, _round_up(K * R, ROW_TILE),
																			   **(dict(into=prev) if prev is not None and isinstance(self.be, HipBackend) else {})))
		self._pd = pd
		dot = self._timed('gram', timed, lambda: self.be.gram(self._blk, pd, False, R, K * R))
		p, stat, self.flags = self._timed('sweep', timed, lambda: self.be.sweep(dot, self._ss, pss, R, K * R, self.n, self.dof, False, self.out_dtype, self.flags,
																				a=self._blk, b=pd))
		for j in range(K):
			outs.append(dict(bi=self.rank, bj=(self.rank + 1 + j) % W, row_lo=0, nx=R, ny=R, symmetric=False,
							 p=p[:, j * R:(j + 1) * R], stat=stat[:, j * R:(j + 1) * R]))

	def _operands(self, bi, bj, lo, hi):
		a, ssa = self.block(bi)
		b, ssb = self.block(bj)
		nx = max(0, min(hi, self.rows) - lo)
		if nx and (lo != 0 or hi != self.rows_pad):
			a, ssa = self.be.rows(a, lo, hi, nx), ssa[lo:hi]
		return a, ssa, b, ssb, nx

	def _step_chunked(self, timed):
		"""One pass with the pipelined exchange: K1 writes this rank's digit planes in cell chunks; every chunk is all-gathered
		by its own collective, queued back to back on RCCL's stream; the own (diagonal) block pair is contracted while the first
		chunk travels, then every other pair of this rank chunk by chunk as the pieces land."""
		import torch.distributed as dist
		be = self.be
		blk, ss = self._timed('residualize', timed, lambda: be.residualize(self.x, self.cov, self.rows_pad, chunks=self.chunks, **self._into()))
		self._blk, self._ss = blk, ss
		S = be.n_chunks(blk)
		chunks, once = be.chunk_payload(blk)
		chunks = [t.contiguous() for t in chunks]
		once = [t.contiguous() for t in once] + [ss.contiguous()]
		fits = lambda gs, ts: gs is not None and len(gs) == len(ts) and all(g.shape[1:] == t.shape and g.dtype == t.dtype for g, t in zip(gs, ts))
		if not (fits(getattr(self, '_g_chunks', None), chunks) and fits(getattr(self, '_g_once', None), once)):
			new = lambda t: t.new_empty((self.world, ) + tuple(t.shape))
			self._g_chunks, self._g_once = [new(t) for t in chunks], [new(t) for t in once]
		nccl = dist.get_backend(self.group) == 'nccl'

		def gather(g, t):
			if nccl:
				return [dist.all_gather_into_tensor(g, t, group=self.group, async_op=True)]
			dist.all_gather(list(g.unbind(0)), t, group=self.group)
			return []

		def wait(handles):
			def w():
				for h in handles:
					h.wait()
			if handles:
				self._timed('exchange', timed, w)  # the part of the flight that compute did not hide
		h_once = [h for g, t in zip(self._g_once, once) for h in gather(g, t)]
		h_chunk = [gather(g, t) for g, t in zip(self._g_chunks, chunks)]
		outs = []

		def finish(e, ops, dot):
			bi, bj, lo, hi, sym = e
			a, ssa, b, ssb, nx = ops
			p, stat, self.flags = self._timed('sweep', timed, lambda: be.sweep(dot, ssa, ssb, nx, self.rows, self.n, self.dof, sym, self.out_dtype, self.flags, a=a, b=b))
			outs.append(dict(bi=bi, bj=bj, row_lo=lo, nx=nx, ny=self.rows, symmetric=sym, p=p, stat=stat))
		own = [e for e in self.sched if e[0] == self.rank and e[1] == self.rank]
		rest = [e for e in self.sched if not (e[0] == self.rank and e[1] == self.rank)]
		if self.forced:  # (one rank standing in for many: its own pair is taken from the gather buffers like a partner's)
			own, rest = [], own + rest
		for e in own:  # local data only: runs while the chunks travel
			ops = self._operands(*e[:4])
			dot = None
			for c in range(S):
				dot = self._timed('gram', timed, lambda: be.gram_chunk(ops[0], ops[2], e[4], c, dot, c > 0))
			finish(e, ops, dot)
		wait(h_once)
		# the full partner blocks rank+1 .. rank+K are consecutive (cyclically) in the gather buffers: ONE launch per chunk for
		# all of them (K x more tiles per launch keep the persistent schedule in its whole-tile regime); what is left -- the
		# half-split pair of an even world -- goes pair by pair
		full = [e for e in rest if e[0] == self.rank and e[2] == 0 and e[3] == self.rows_pad and not e[4]]
		K = len(full)
		merged = K >= 2 and hasattr(be, 'gram_chunk_blocks') and _opts.debug('merge_partners', '1') != '0' and [
			e[1] for e in full] == [(self.rank + 1 + j) % self.world for j in range(K)]
		if merged:
			rest = [e for e in rest if e not in full]
		todo = [(e, self._operands(*e[:4])) for e in rest]
		todo = [(e, ops) for e, ops in todo if ops[4] > 0]
		dots = [None] * len(todo)
		mdot = None
		for c in range(S):
			wait(h_chunk[c])
			if merged:
				mdot = self._timed('gram', timed, lambda: be.gram_chunk_blocks(blk, self._g_chunks[c], self._g_once, (self.rank + 1) % self.world, K, c,
																			   mdot, c > 0))
			for i, (e, ops) in enumerate(todo):
				dots[i] = self._timed('gram', timed, lambda: be.gram_chunk(ops[0], ops[2], e[4], c, dots[i], c > 0))
		if merged:
			for j, e in enumerate(full):
				ssb = self._g_once[-1][e[1]]
				finish(e, (blk, ss, self.block(e[1])[0], ssb, self.rows), mdot[:, j * self.rows_pad:(j + 1) * self.rows_pad])
		for (e, ops), dot in zip(todo, dots):
			finish(e, ops, dot)
		self.outputs = outs
		return outs

	@_on_engine
	def step(self, timed=False):
		if timed:
			self._timed_steps += 1
		self._pending = []
		self._partner = {}
		if self.multi and self.chunks:
			return self._step_chunked(timed)
		if self.multi and self.exchange_raw:
			self._pending = self._exchange(None, None)  # raw rows travel: nothing to wait for, start before K1
		blk, ss = self._timed('residualize', timed, lambda: self.be.residualize(self.x, self.cov, self.rows_pad, **self._into()))
		self._blk, self._ss = blk, ss
		if self.multi and not self.exchange_raw:
			self._pending = self._exchange(blk, ss)
		outs = []
		merged = self.exchange_raw and self.n_partners >= 1 and _opts.debug('merge_partners', '1') != '0'
		merged_done = False
		for bi, bj, lo, hi, sym in self.sched:
			if self._pending and (self.forced or not (bi == self.rank and bj == self.rank)):
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

	# ---- complete row blocks on the owner, without rank 0 in the middle ---------------------------------------
	def _mirror_messages(self):
		"""Every off-diagonal schedule entry (rank s computed rows [lo, hi) of block bi against block bj) as seen from this
		rank: the owner of block bi needs those rows as they are, the owner of block bj needs them transposed.
		Returns (local, sends, recvs): lists of (entry, transposed) / (peer, entry, transposed) in one global order."""
		local, sends, recvs = [], [], []
		for s_rank in range(self.world):
			for bi, bj, lo, hi, sym in block_pair_schedule(s_rank, self.world, self.rows_pad):
				if sym:
					continue
				hi_v = min(hi, self.rows)
				if hi_v <= lo:
					continue
				e = (bi, bj, lo, hi_v)
				for owner, transposed in ((bi, False), (bj, True)):
					if s_rank == self.rank and owner == self.rank:
						local.append((e, transposed))
					elif s_rank == self.rank:
						sends.append((owner, e, transposed))
					elif owner == self.rank:
						recvs.append((s_rank, e, transposed))
		return local, sends, recvs

	@_on_engine
	def complete_rows(self):
		"""After step(): the complete rows of this rank's gene block, (rows, world * rows) for p and for the covariance, on
		this rank's device.  The blocks this rank did not compute itself arrive from the ranks that did (mirrored blocks
		transposed by the sender) in ONE point-to-point exchange -- every block travels to exactly the one rank that needs
		it, nothing passes through rank 0 and nothing is pickled.  (association.py:1049-1057 mirrors the upper triangle in
		the same way, on one host.)"""
		torch = self.be.torch
		R, W = self.rows, self.world
		as_t = lambda v: v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))
		outs = {(o['bi'], o['bj'], o['row_lo']): (as_t(o['p']), as_t(o['stat'])) for o in self.outputs}
		own = outs[(self.rank, self.rank, 0)]
		P = torch.empty((R, W * R), dtype=own[0].dtype, device=own[0].device)
		S = torch.empty((R, W * R), dtype=own[1].dtype, device=own[1].device)
		P[:, self.rank * R:(self.rank + 1) * R] = own[0]
		S[:, self.rank * R:(self.rank + 1) * R] = own[1]

		def place(e, transposed, p, st):
			bi, bj, lo, hi = e
			if transposed:  # rows of block bj (all of them), columns lo..hi of block bi
				P[:, bi * R + lo:bi * R + hi] = p
				S[:, bi * R + lo:bi * R + hi] = st
			else:
				P[lo:hi, bj * R:(bj + 1) * R] = p
				S[lo:hi, bj * R:(bj + 1) * R] = st
		local, sends, recvs = self._mirror_messages()
		for e, transposed in local:
			p, st = outs[(e[0], e[1], e[2])]
			place(e, transposed, p.t() if transposed else p, st.t() if transposed else st)
		if W > 1:
			import torch.distributed as dist
			on_host = dist.get_backend(self.group) != 'nccl'  # gloo moves host tensors only (functional tests); RCCL sends from HBM
			ops, inbox, keep = [], [], []
			for peer, e, transposed in sends:
				p, st = outs[(e[0], e[1], e[2])]
				msg = torch.stack([p.t() if transposed else p, st.t() if transposed else st]).contiguous()
				msg = msg.cpu() if on_host else msg
				keep.append(msg)
				ops.append(dist.P2POp(dist.isend, msg, peer, group=self.group))
			for peer, e, transposed in recvs:
				shape = (2, R, e[3] - e[2]) if transposed else (2, e[3] - e[2], R)
				buf = torch.empty(shape, dtype=P.dtype, device='cpu' if on_host else P.device)
				inbox.append((e, transposed, buf))
				ops.append(dist.P2POp(dist.irecv, buf, peer, group=self.group))
			if ops:
				for w in dist.batch_isend_irecv(ops):
					w.wait()
			for e, transposed, buf in inbox:
				buf = buf.to(P.device)
				place(e, transposed, buf[0], buf[1])
		return P, S

	def _flags_ok_everywhere(self):
		"""The reference's assertions (association.py:248,252) as ONE decision of all ranks: a rank that raised alone
		would leave the others waiting in the next collective."""
		if self.flags is None:
			return
		f = self.flags
		if self.world > 1:
			f = _reduce_flags(f, self.group)
		self.be.eng.check_flags(f)

	@_on_engine
	def binnet_rows(self, p_rows, qcut):
		"""binnet (binnet.py:134-173) of this rank's complete row block, in HBM: per-row BH q-values need nothing but the
		row.  Returns the (rows, n_gene) uint8 block; the "Empty binary network" test is made on the sum over ranks."""
		from . import _lib
		eng = self.be.eng
		torch = eng.torch
		R, ng = p_rows.shape
		with torch.cuda.device(eng.device):
			out = torch.empty((R, ng), dtype=torch.uint8, device=eng.device)
			total = torch.zeros(1, dtype=torch.int64, device=eng.device)
			flags = torch.zeros(2, dtype=torch.int32, device=eng.device)
			_lib.check(eng.lib.nrm_binnet_rows(p_rows.data_ptr(), _lib.NRM_F64 if p_rows.dtype == torch.float64 else _lib.NRM_F32, R, ng,
											   p_rows.stride(0), self.rank * R, float(qcut), out.data_ptr(), out.stride(0), total.data_ptr(),
											   flags.data_ptr(), eng._stream()))
			stats = torch.stack([total[0], flags[0].to(torch.int64)])
			if self.world > 1:
				import torch.distributed as dist
				stats = stats if dist.get_backend(self.group) == 'nccl' else stats.cpu()
				dist.all_reduce(stats, group=self.group)
			bad, tot = int(stats[1].item()), int(stats[0].item())
		if bad:
			raise AssertionError('P-values must be finite and within [0,1] (binnet.py:151-152).')
		if tot == 0:
			raise RuntimeError('Empty binary network.')
		return out

	def _to_host_rows(self, t, dst):
		"""This rank's row block -> its rows of a result array shared by the ranks."""
		if hasattr(self.be, 'eng') and t.is_cuda:
			self.be.eng.download_into(t, dst)
		else:
			dst[...] = t.numpy() if hasattr(t, 'numpy') else np.asarray(t)

	@_on_engine
	def assemble(self, to_numpy=None, out_dir=None):
		"""(p, dot, var) of the whole problem on rank 0 (None elsewhere) with the reference's contract (symmetric, zero
		diagonals, coex.py:4-48): every rank completes its row block on its device and copies it into ITS rows of result
		arrays shared by the ranks (SharedArrays).  out_dir: keep the arrays as p.npy / dot.npy / var.npy there."""
		P, S = self.complete_rows()
		R, ng = self.rows, self.world * self.rows
		sh = SharedArrays(dict(p=((ng, ng), self.out_dtype), dot=((ng, ng), self.out_dtype), var=((ng, ), self.out_dtype)),
						  self.rank, self.world, self.group, base=out_dir, keep=out_dir is not None)
		a, b = self.rank * R, (self.rank + 1) * R
		self._to_host_rows(P, sh['p'][a:b])
		self._to_host_rows(S, sh['dot'][a:b])
		ss = self._ss[:R]
		var = (ss.cpu().numpy() if hasattr(ss, 'cpu') else np.asarray(ss)) / float(self.n)
		var[var == 0] = 1
		sh['var'][a:b] = var
		res = sh.finish()
		return None if res is None else (res['p'], res['dot'], res['var'])


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
		self._graph = StepGraph(self.eng.torch)
		# buffers of the streaming path that a captured graph points into, and the row scales of the resident expression rows: owned by
		# the plan (see association_de_streaming); keep=True: the plan is resident, its rows will be streamed again
		self._state = {'keep': True}

	def _run(self):
		# resident step: p / gamma / sums of squares stay in HBM; results() brings them to the host and checks the flags
		return self.eng.association_single0(self.dx, self.dy, self.dc64, self.dci, self.dcr, self.dimreduce,
											return_dot=self.return_dot, want_alpha=False, out_dtype=self.out_dtype, cov=self.cov,
											resident=True, state=self._state)

	@_on_engine
	def step(self, timed=False):
		torch = self.eng.torch
		if timed:
			e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			e0.record()
		self.result = self._graph.run(self._run)
		if timed:
			e1.record()
			self._ev.append((e0, e1))
		return self.result

	@_on_engine
	def results(self):
		"""(p, gamma, varx, vary) of the last step as numpy arrays, after the reference's assertions (association.py:248,252)."""
		from .engine import GuardHit
		r = self.result
		try:
			self.eng.check_flags(r['flags'])
		except GuardHit as g:
			st = self._state
			sp = st.get('sparse')
			if sp is not None and sp[0] is self.dx and sp[2].ok and st.get('sparse_refused') is not self.dx:
				# The sparse-design kernels handed the step back: rows (expression or design) all but inside the span of the covariates, for
				# which their differences lose digits.  That is a property of the data, not of the step: the plan remembers it and takes K1 +
				# the Gram engines from now on (captured as a graph again) instead of trying the sparse kernels first at every step.
				import logging
				logging.warning('normalisr_amd: %d rows too close to the span of the covariates for the sparse-design kernels; this plan runs on K1 and the Gram engines from now on.', g.hits)
				st['sparse_refused'] = self.dx
				self._graph.graph, self._graph.calls = None, 0
				try:
					r = self.result = self._run()
					self.eng.check_flags(r['flags'])
					g = None
				except GuardHit as g2:
					g = g2
			if g is not None:  # the integer engine could not certify every P-value: this step again, eagerly, on the fp64 Gram kernel
				st['keep'] = False
				self._graph.enabled, self._graph.graph = False, None
				with self.eng.forced_f64():
					r = self._run()
					self.eng.check_flags(r['flags'])
				self.eng.last_guard = dict(hits=g.hits, worst=g.worst, fallback=True)
		return (self.eng.download(r['p']), self.eng.download(r['stat']), self.eng.variances(r['ssx'], self.nx, self.n, self.out_dtype),
				self.eng.variances(r['ssy'], self.ny, self.n, self.out_dtype))

	def step_ms(self):
		self.eng.torch.cuda.synchronize()
		return sum(a.elapsed_time(b) for a, b in self._ev) / max(1, len(self._ev))

	def streaming(self):
		return self.eng.de_streaming_ok(self.dx, self.dy, self.dc64)


def _local_rows(dt_local, dev):
	import torch
	if isinstance(dt_local, np.ndarray):
		a = dt_local if dt_local.dtype in (np.float32, np.float64) else dt_local.astype(np.float64)
		dt_local = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
	return dt_local


def _world(group):
	import torch.distributed as dist
	if not dist.is_initialized():
		return 0, 1, None
	return dist.get_rank(group), dist.get_world_size(group), (group if group is not None else dist.group.WORLD)


def _certified_coex_step(x, dc, rank, world, group, dimreduce):
	"""One sharded coex pass whose P-values are certified: the reference's assertions and the integer engine's accuracy guard are
	ONE decision of all ranks (the device counters are reduced over the group), and a pass the guard cannot certify is redone by
	every rank on the fp64 Gram kernel."""
	from .engine import GuardHit
	plan = CoexPlan(x, dc, rank=rank, world=world, group=group, dimreduce=dimreduce)
	plan.step()
	try:
		plan._flags_ok_everywhere()
	except GuardHit as g:
		eng = plan.be.eng
		with eng.forced_f64():
			plan = CoexPlan(x, dc, rank=rank, world=world, group=group, dimreduce=dimreduce)
			plan.step()
			plan._flags_ok_everywhere()
		eng.last_guard = dict(hits=g.hits, worst=g.worst, fallback=True)
	return plan


def coex(dt_local, dc, group=None, dimreduce=0, out_dir=None):
	"""Sharded norm.coex for one-process-per-GPU programs: every rank passes ITS block of gene rows (same row count on
	every rank; numpy or a torch tensor on its GPU) and the replicated covariates.  Rank 0 gets (p, dot, var) as numpy
	arrays with the reference's contract (symmetric, zero diagonals; coex.py:4-48); other ranks get None.  Every rank
	copies its own rows into arrays shared by the ranks (see CoexPlan.assemble); out_dir keeps them as .npy files.

	    # torchrun --nproc-per-node 8 script.py
	    dist.init_process_group('nccl'); torch.cuda.set_device(local_rank)
	    res = normalisr_amd.distributed.coex(dt[rank * R:(rank + 1) * R], dc)
	"""
	import torch
	rank, world, group = _world(group)
	dev = torch.device('cuda', torch.cuda.current_device())
	plan = _certified_coex_step(_local_rows(dt_local, dev), dc, rank, world, group, dimreduce)
	return plan.assemble(out_dir=out_dir)


def coex_binnet(dt_local, dc, qcut, group=None, dimreduce=0, out_dir=None):
	"""coex -> binnet as one sharded device pipeline (examples/GSE123139/code/cmd_coex.sh:40-42 chains the two through a
	P-value file): every rank completes its rows of the P-value matrix in HBM and binarises them there (per-row BH,
	binnet.py:134-173), so the n_gene^2 P-values never cross PCIe -- only one byte per pair does.  Rank 0 gets the boolean
	(n_gene, n_gene) network (net.npy in out_dir if given), other ranks None."""
	import torch
	rank, world, group = _world(group)
	dev = torch.device('cuda', torch.cuda.current_device())
	plan = _certified_coex_step(_local_rows(dt_local, dev), dc, rank, world, group, dimreduce)
	P, _ = plan.complete_rows()
	net = plan.binnet_rows(P, qcut)
	R, ng = plan.rows, world * plan.rows
	sh = SharedArrays(dict(net=((ng, ng), np.bool_)), rank, world, group, base=out_dir, keep=out_dir is not None)
	plan._to_host_rows(net, sh['net'][rank * R:(rank + 1) * R].view(np.uint8))  # the kernel writes exact 0/1 bytes
	res = sh.finish()
	return None if res is None else res['net']


def _de_other_methods(dg, dt_local, dc, group, dimreduce, out_dir, single, ka):
	"""Sharded de for single=1 (every grouping on the cells free of the other groupings, association.py:263-390) and single=4 (the
	other groupings as covariates, :421-576): a gene's results depend on the design, the covariates and that gene's row only, so
	every rank runs the single-GPU path on ITS gene rows and writes its columns of the shared result arrays; the small design-side
	factorisations are repeated per rank (replicas of the (n_x + n_cov)^2 inverse + gene-row sharding)."""
	from .de import de as de_local
	rank, world, group = _world(group)
	y = dt_local.detach().cpu().numpy() if hasattr(dt_local, 'detach') else np.asarray(dt_local)
	dimr = dimreduce
	counts = np.array([[y.shape[0]]])
	if world > 1:
		import torch
		counts = _all_gather_ints([y.shape[0]], group, torch.device('cuda', torch.cuda.current_device()))
	starts = np.concatenate([[0], np.cumsum(counts[:, 0])])
	a, b = int(starts[rank]), int(starts[rank + 1])
	if np.ndim(dimr) != 0:  # one value per gene: this rank's genes
		dimr = np.asarray(dimr)[a:b]
	err, res = None, None
	try:
		res = de_local(dg, y, dc, single=single, dimreduce=dimr, **ka)
	except Exception as e:  # every rank must learn of it: the others would wait in the collectives below
		err = e
	if world > 1:
		import torch.distributed as dist
		bad = _coll_tensor([0 if err is None else 1], group, torch.device('cuda', torch.cuda.current_device()))
		dist.all_reduce(bad, group=group)
		if int(bad.item()) and err is None:
			err = RuntimeError('sharded de (single={}) failed on another rank'.format(single))
	if err is not None:
		raise err
	p, gam, alpha, vg, vt = res
	ng0, nt = p.shape[0], int(starts[-1])
	odt = p.dtype
	sh = SharedArrays(dict(p=((ng0, nt), odt), gamma=((ng0, nt), odt), varg=((ng0, ), odt), vart=((ng0, nt), odt)), rank, world, group,
					  base=out_dir, keep=out_dir is not None)
	sh['p'][:, a:b], sh['gamma'][:, a:b], sh['vart'][:, a:b] = p, gam, vt
	if rank == 0:
		sh['varg'][...] = vg
	out = sh.finish()
	return None if out is None else (out['p'], out['gamma'], None, out['varg'], out['vart'])


def de(dg, dt_local, dc, group=None, dimreduce=0, out_dir=None, single=0, **ka):
	"""Sharded norm.de for one-process-per-GPU programs: every rank passes the full grouping matrix dg, ITS
	block of gene rows of dt (numpy or a torch tensor on its GPU; the blocks may differ in size) and the replicated
	covariates.  No collective on the data path; every rank writes its gene columns into result arrays shared by the ranks
	and rank 0 returns (p, gamma, None, varg, vart) with the reference's contract (de.py:4-132, constant groupings
	re-inflated), other ranks get None.  single = 0 (resident DePlan), 1 or 4 (see _de_other_methods)."""
	if single not in (0, 1, 4):
		raise ValueError('Unknown value single={}'.format(single))
	if single:
		return _de_other_methods(dg, dt_local, dc, group, dimreduce, out_dir, single, ka)
	if ka:
		raise TypeError("de() got an unexpected keyword argument '{}'".format(next(iter(ka))))
	import torch
	from .de import _varying_rows
	rank, world, group = _world(group)
	dev = torch.device('cuda', torch.cuda.current_device())
	dg = np.asarray(dg)
	gid = _varying_rows(dg)  # de.py:93
	dt_local = _local_rows(dt_local, dev)
	x = dg[gid]
	x = x if x.dtype in (np.float32, np.float64) else x.astype(np.float64)
	plan = DePlan(torch.from_numpy(np.ascontiguousarray(x)).to(dev), dt_local, dc, rank=rank, world=world, dimreduce=dimreduce)
	plan.step()
	flags = plan.result['flags']
	counts = np.array([[plan.ny]])
	if world > 1:
		import torch.distributed as dist
		plan.result['flags'] = _reduce_flags(flags, group)  # one decision for all ranks (see CoexPlan._flags_ok_everywhere)
		counts = _all_gather_ints([plan.ny], group, dev)
	p, gam, vg, vt = plan.results()
	starts = np.concatenate([[0], np.cumsum(counts[:, 0])])
	ng0, nt = dg.shape[0], int(starts[-1])
	odt = plan.out_dtype
	sh = SharedArrays(dict(p=((ng0, nt), odt), gamma=((ng0, nt), odt), varg=((ng0, ), odt), vart=((ng0, nt), odt)), rank, world, group,
					  base=out_dir, keep=out_dir is not None)
	a, b = int(starts[rank]), int(starts[rank + 1])
	# de.py:107-122: tested groupings get their results, constant ones p = 1, gamma = 0, variances 0
	P, G, VT = sh['p'], sh['gamma'], sh['vart']
	P[:, a:b], G[:, a:b], VT[:, a:b] = 1, 0, 0
	P[gid, a:b], G[gid, a:b], VT[gid, a:b] = p, gam, vt
	if rank == 0:
		sh['varg'][...] = 0
		sh['varg'][gid] = vg
	res = sh.finish()
	return None if res is None else (res['p'], res['gamma'], None, res['varg'], res['vart'])
