"""One rank of `normalisr coex|de --gpus N` (started by normalisr_amd.launch; never run by hand).

Shard at load: the expression matrix is never read whole by any rank -- a .npy file is memory-mapped and only this rank's rows
are copied out of it, a TSV is read from this rank's first line for this rank's line count.  coex needs the same number of rows
on every rank: the last block is padded with zero rows (variance 0 -> 1, P = 1 like the reference's zero-variance rows,
association.py:231-233) and the padding is cropped before the files are written.
"""
import json
import logging
import os
import sys

import numpy as np


def _tsv_data_lines(path):
	"""Byte offsets (start, end) of the lines of a TSV that np.loadtxt counts as rows: not blank, not a '#' comment (loadtxt drops
	both and counts neither towards skiprows / max_rows, so raw line numbers would shift the ranks' blocks).  One scan, no parsing."""
	import gzip
	op = gzip.open if path.endswith('.gz') else open
	spans, pos = [], 0
	with op(path, 'rb') as f:
		for line in f:
			body = line.split(b'#', 1)[0].strip()
			if body:
				spans.append((pos, pos + len(line)))
			pos += len(line)
	return spans


def matrix_rows(path):
	"""Row count of a matrix file without reading it: the .npy header, or the number of data lines of a TSV."""
	if path.endswith('.npy'):
		a = np.load(path, mmap_mode='r', allow_pickle=False)
		return 1 if a.ndim < 2 else a.shape[0]
	return len(_tsv_data_lines(path))


def read_rows(path, lo, hi, spans=None):
	"""Rows [lo, hi) of a matrix file, 2-D, without reading the others into memory (TSV: only this block's data lines are handed
	to np.loadtxt).  An empty block comes back as shape (0, 0): the caller knows the column count."""
	if path.endswith('.npy'):
		a = np.load(path, mmap_mode='r', allow_pickle=False)
		a = a.reshape(1, -1) if a.ndim < 2 else a
		return np.ascontiguousarray(a[lo:hi])
	if hi <= lo:
		return np.zeros((0, 0))
	import gzip
	import io
	spans = _tsv_data_lines(path) if spans is None else spans
	op = gzip.open if path.endswith('.gz') else open
	with op(path, 'rb') as f:
		f.seek(spans[lo][0])
		blob = f.read(spans[hi - 1][1] - spans[lo][0])  # (comment / blank lines inside the block are dropped by the parser itself)
	from . import run
	if run._native_text():
		out = run.parse_text(np.frombuffer(blob, dtype=np.uint8))
		if out is not None:
			return out
	return np.loadtxt(io.BytesIO(blob), delimiter='\t', ndmin=2)


def block_bounds(rows, world, rank, balanced):
	"""Gene rows [lo, hi) of `rank`.  coex needs equal blocks (ceil(rows / world), the tail padded by the caller: the block pair
	schedule works on one block size); de takes balanced blocks, so that no rank is left without rows while rows >= world."""
	if balanced:
		base, extra = divmod(rows, world)
		lo = rank * base + min(rank, extra)
		return lo, lo + base + (1 if rank < extra else 0)
	per = -(-rows // world)
	return min(rows, rank * per), min(rows, (rank + 1) * per)


def main(argv):
	job = json.loads(argv[0])
	cmd, args = job['cmd'], job['args']
	rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
	logging.basicConfig(format='%(levelname)s:rank {}:%(asctime)s:%(message)s'.format(rank), level=logging.DEBUG if args.get('verbose') else logging.WARNING)
	import torch
	import torch.distributed as dist
	from . import distributed as nd
	from .run import file_read_tsv, file_write_tsv, fmt_float
	local = 0 if os.environ.get('NRM_SHARE_GPU') == '1' else int(os.environ.get('LOCAL_RANK', rank))
	torch.cuda.set_device(local)
	backend = os.environ.get('NRM_DIST_BACKEND', 'nccl')
	if backend == 'nccl':
		dist.init_process_group('nccl', device_id=torch.device('cuda', local))
	else:
		dist.init_process_group(backend)
	try:
		dimr = int(args['dimr']) if args.get('dimr') is not None else 0
		dc = file_read_tsv(args['cov_in'])
		rows = matrix_rows(args['exp_in'])
		if rows < world:
			raise ValueError('{} gene rows cannot be sharded over {} GPUs'.format(rows, world))
		if cmd == 'coex':
			per = -(-rows // world)
			lo, hi = block_bounds(rows, world, rank, False)
			x = read_rows(args['exp_in'], lo, hi)
			if x.dtype not in (np.float32, np.float64):
				x = x.astype(np.float64)
			if x.shape[0] == 0:  # a tail rank without rows of its own (e.g. 9 genes on 8 GPUs): all padding
				x = np.zeros((0, dc.shape[1]), dtype=x.dtype)
			if x.shape[0] < per:  # the last block(s): zero rows up to the common block size
				x = np.vstack([x, np.zeros((per - x.shape[0], dc.shape[1]), dtype=x.dtype)])
			logging.debug('rows %d..%d of %d loaded', lo, hi, rows)
			res = nd.coex(x, dc, dimreduce=dimr)
			if rank == 0:
				p, dot, var = (np.asarray(a)[:rows] for a in res)
				out = dict(pv_out=p[:, :rows], dot_out=dot[:, :rows], var_out=var)
		else:
			lo, hi = block_bounds(rows, world, rank, True)
			dg = file_read_tsv(args['design_in'])
			y = read_rows(args['exp_in'], lo, hi)
			if y.dtype not in (np.float32, np.float64):
				y = y.astype(np.float64)
			logging.debug('gene rows %d..%d of %d loaded', lo, hi, rows)
			from .run import _de_method
			res = nd.de(dg, y, dc, dimreduce=dimr, single=_de_method(args.get('method') or 'ignore'))
			if rank == 0:
				out = dict(pv_out=res[0], lfc_out=res[1], vard_out=res[3], vart_out=res[4])
		if rank == 0:
			for key, val in out.items():
				if args.get(key) is not None:
					file_write_tsv(args[key], val, fmt=fmt_float)
		dist.barrier()
	except BaseException:
		# no orderly teardown on the failure path: destroy_process_group() can block on collectives the peers will never match, the
		# parent would then never see this rank exit and the survivors would sit in their collective for ever
		import traceback
		traceback.print_exc()
		sys.stderr.flush()
		os._exit(1)
	dist.destroy_process_group()
	return 0


if __name__ == '__main__':
	sys.exit(main(sys.argv[1:]))
