"""Command runners and matrix IO behind `normalisr de | coex | binnet | normvar`.

Every sub-command is one row of COMMANDS: which files are read (and how they are shaped), which command-line
options become which keyword arguments, which function runs, and which of its results go to which file.
The file contract is the reference's (run.py:20-35,258-321): tab-delimited text without headers, one row
per line, '%.8G' for floats, '.gz' by suffix (numpy handles it), a single row read back as shape (1, n).
Extension of this build: names ending in '.npy' are read / written as binary numpy arrays; and the text files themselves go
through the library's threaded parser / printer (csrc/nrm_tsv.hip: the same numbers in, byte for byte the same text out) instead of
numpy.loadtxt / numpy.savetxt, which take minutes for a 20k x 100k matrix where the association takes milliseconds (NRM_TSV=numpy: back).
"""
import logging

import numpy as np

from . import _opts

fmt_float = '%.8G'
fmt_int = '%i'


def _is_binary(name):
	return name.endswith('.npy')


def _native_text():
	"""The library's text reader / writer unless NRM_TSV=numpy asks for numpy.loadtxt / numpy.savetxt (the reference's own calls)."""
	import os
	return _opts.debug('tsv', 'native') != 'numpy'


def _open_bytes(f):
	if f.endswith('.gz'):  # '.gz' by suffix, as numpy does for the reference
		import gzip
		with gzip.open(f, 'rb') as fh:
			return np.frombuffer(fh.read(), dtype=np.uint8)
	if f.endswith('.bz2') or f.endswith('.xz'):
		return None
	return np.fromfile(f, dtype=np.uint8)


def parse_text(buf, delimiter='\t', dtype=np.float64):
	"""The matrix in a buffer of text (uint8 array), 2-D, through the library's threaded parser (csrc/nrm_tsv.hip): the numbers
	numpy.loadtxt reads (correctly rounded, as float()), '#' comments and blank lines skipped, ValueError for a field that is not a number
	or a row of another length.  None for a buffer without data (numpy warns and returns an empty array: left to numpy)."""
	import ctypes
	from . import _lib
	lib = _lib.load()
	buf = np.ascontiguousarray(buf, dtype=np.uint8)
	rows, cols = ctypes.c_int64(), ctypes.c_int64()
	_lib.check(lib.nrm_tsv_shape(buf.ctypes.data, buf.size, ord(delimiter), 0, ctypes.addressof(rows), ctypes.addressof(cols)))
	if rows.value == 0:
		return None
	out = np.empty((rows.value, cols.value), dtype=dtype)
	_lib.check(lib.nrm_tsv_parse(buf.ctypes.data, buf.size, ord(delimiter), 0, out.ctypes.data, _lib.NRM_F64 if out.dtype == np.float64 else _lib.NRM_F32,
								 rows.value, cols.value, cols.value))
	return out


def _read_text(f, delimiter, dtype):
	"""numpy.loadtxt(f, delimiter=delimiter) through parse_text."""
	buf = _open_bytes(f)
	if buf is None:
		return None
	try:
		out = parse_text(buf, delimiter, dtype)
	except ValueError:
		return None  # text the library's parser does not take: numpy.loadtxt answers -- its matrix or its exception are the reference's (run.py:20-27)
	return None if out is None else out.squeeze()  # loadtxt's own squeeze (a single row or column comes back 1-D)


def file_read_tsv(f, delimiter='\t', **ka):
	"""Matrix from a TSV (or .npy) file, always 2-D."""
	logging.debug('Start reading file ' + f)
	ans = None
	if _is_binary(f):
		ans = np.load(f, allow_pickle=False)
		if 'dtype' in ka:
			ans = ans.astype(ka['dtype'], copy=False)
	elif _native_text() and len(delimiter) == 1 and set(ka) <= {'dtype'} and np.dtype(ka.get('dtype', np.float64)) in (np.dtype(np.float32), np.dtype(np.float64)):
		ans = _read_text(f, delimiter, np.dtype(ka.get('dtype', np.float64)))
	if ans is None:
		ans = np.loadtxt(f, delimiter=delimiter, **ka)
	logging.debug('Finish reading file ' + f)
	return ans.reshape(1, -1) if ans.ndim < 2 else ans


def _write_text(f, d, delimiter, fmt):
	"""numpy.savetxt(f, d, delimiter=delimiter, fmt=fmt) for the two formats the command line writes ('%.8G' of floats, '%i' of integers),
	printed by the library's threads in blocks of rows; byte for byte the text numpy writes.  False: not a case of ours."""
	import os
	from . import _lib
	d = np.asarray(d)
	if d.ndim == 1:
		d = d.reshape(-1, 1)  # savetxt writes a vector as a column
	if d.ndim != 2 or len(delimiter) != 1 or f.endswith('.bz2') or f.endswith('.xz'):
		return False
	if fmt == fmt_float and d.dtype in (np.float32, np.float64):
		kind, code = 0, _lib.NRM_F64 if d.dtype == np.float64 else _lib.NRM_F32
	elif fmt == fmt_int and d.dtype.kind in 'biu' and d.dtype != np.uint64:
		kind = 1
		if d.dtype.itemsize == 1 and d.dtype.kind in 'bu':
			d, code = d.view(np.uint8), _lib.NRM_TSV_U8
		elif d.dtype == np.int32:
			code = _lib.NRM_TSV_I32
		else:
			d, code = d.astype(np.int64), _lib.NRM_TSV_I64
	else:
		return False
	lib = _lib.load()
	d = np.ascontiguousarray(d)
	rows, cols = d.shape
	width = int(lib.nrm_tsv_width(kind)) * max(cols, 1)
	block = max(1, min(rows, (256 << 20) // width))  # rows per call: at most 256 MB of text at a time
	parts = max(1, min(os.cpu_count() or 1, 64, block))
	cap = -(-block // parts) * width
	text = np.empty(parts * cap, dtype=np.uint8)
	lens = np.zeros(parts, dtype=np.int64)
	if f.endswith('.gz'):
		import gzip
		fh = gzip.open(f, 'wb')
	else:
		fh = open(f, 'wb')
	with fh:
		for r0 in range(0, rows, block):
			r1 = min(rows, r0 + block)
			_lib.check(lib.nrm_tsv_format(d[r0:r1].ctypes.data, code, r1 - r0, cols, cols, ord(delimiter), kind, text.ctypes.data, cap, lens.ctypes.data, parts))
			for t in range(parts):
				if lens[t]:
					fh.write(memoryview(text[t * cap:t * cap + int(lens[t])]))
	return True


def file_write_tsv(f, d, delimiter='\t', fmt=fmt_float, **ka):
	"""Matrix or vector to a TSV (or .npy) file."""
	logging.debug('Start writing file ' + f)
	if _is_binary(f):
		np.save(f, np.asarray(d), allow_pickle=False)
	elif not (_native_text() and not ka and _write_text(f, d, delimiter, fmt)):
		np.savetxt(f, d, delimiter=delimiter, fmt=fmt, **ka)
	logging.debug('Finish writing file ' + f)


_DE_METHODS = {'ignore': 0, 'single': 1, 'covariate': 4}


def _de_method(name):
	if name not in _DE_METHODS:
		raise ValueError('Unknown method {}'.format(name))
	return _DE_METHODS[name]


def _flat_alpha(a):
	return a.reshape(a.shape[0], -1)  # (predictor, gene * covariate), row-major


def _call_de(m, ka):
	from .de import de
	return de(m['design_in'], m['exp_in'], m['cov_in'], **ka)


def _call_coex(m, ka):
	from .coex import coex
	return coex(m['exp_in'], m['cov_in'], **ka)


def _call_normvar(m, ka):
	from .norm import normvar
	return normvar(m['lcpm_in'], m['cov_in'], m['weights_in'].ravel(), m['scale_in'].ravel(), **ka)


def _call_binnet(m, ka):
	from .binnet import binnet
	return (binnet(m['pv_in'], ka['qcut']).astype('u1', copy=False), )


# name -> inputs (matrix arguments), options (argument key -> (keyword, converter)), call, outputs (argument key ->
# (index into the result tuple, transform, format); written when the argument was given)
COMMANDS = {
	'de': dict(inputs=('design_in', 'exp_in', 'cov_in'),
			   options=dict(nth=('nth', int), bs=('bs', int), dimr=('dimreduce', int), method=('single', _de_method),
							# the reference leaves lowmem=True and then fails writing None: asking for the file asks for alpha
							clfc_out=('lowmem', lambda name: False)),
			   call=_call_de,
			   outputs=dict(pv_out=(0, None, fmt_float), lfc_out=(1, None, fmt_float), clfc_out=(2, _flat_alpha, fmt_float),
							vard_out=(3, None, fmt_float), vart_out=(4, None, fmt_float))),
	'coex': dict(inputs=('exp_in', 'cov_in'), options=dict(nth=('nth', int), bs=('bs', int), dimr=('dimreduce', int)), call=_call_coex,
				 outputs=dict(pv_out=(0, None, fmt_float), dot_out=(1, None, fmt_float), var_out=(2, None, fmt_float))),
	'normvar': dict(inputs=('lcpm_in', 'cov_in', 'weights_in', 'scale_in'), options=dict(nth=('nth', int), bs=('bs', int)), call=_call_normvar,
					outputs=dict(exp_out=(0, None, fmt_float), cov_out=(1, None, fmt_float))),
	'binnet': dict(inputs=('pv_in', ), options=dict(qcut=('qcut', float)), call=_call_binnet, outputs=dict(net_out=(0, None, fmt_int))),
}


def run(cmd, args):
	"""Run sub-command `cmd` with the parsed command line `args` (a dict of argparse destinations)."""
	spec = COMMANDS[cmd]
	if cmd in ('de', 'coex', 'binnet', 'normvar'):
		# files in, files out, one GPU: the library's whole-problem entries (include/normalisr_hip.h) do everything these commands need -- same kernels,
		# same results -- and torch's import (1.0 of a 1.3 s call) is not paid; NRM_HOST_ENTRY=0 keeps the torch engine.  Calls the entries do not
		# cover fall back to it by themselves.
		from . import _lib
		prev = _lib.prefer_host_entry(True)
		try:
			return _run(cmd, spec, args)
		finally:
			_lib.prefer_host_entry(prev)  # (a preference of this call, not of the process: tests and notebooks call run() beside the torch engine)
	return _run(cmd, spec, args)


def _run(cmd, spec, args):
	mats = {k: file_read_tsv(args[k]) for k in spec['inputs']}
	ka = {}
	for key, (kw, conv) in spec['options'].items():
		if args.get(key) is not None:
			ka[kw] = conv(args[key])
	logging.debug('Start calculation.')
	res = spec['call'](mats, ka)
	logging.debug('Finish calculation.')
	for key, (idx, transform, fmt) in spec['outputs'].items():
		if args.get(key) is not None:
			file_write_tsv(args[key], res[idx] if transform is None else transform(res[idx]), fmt=fmt)


def _runner(cmd):
	def f(args):
		return run(cmd, args)
	f.__name__ = cmd
	f.__doc__ = 'normalisr {} (see COMMANDS)'.format(cmd)
	return f


de, coex, normvar, binnet = (_runner(c) for c in ('de', 'coex', 'normvar', 'binnet'))  # module-level entry points, as in the reference's run module

assert __name__ != "__main__"
