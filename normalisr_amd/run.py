"""Command runners and matrix IO behind `normalisr de | coex | binnet | normvar`.

Every sub-command is one row of COMMANDS: which files are read (and how they are shaped), which command-line
options become which keyword arguments, which function runs, and which of its results go to which file.
The file contract is the reference's (run.py:20-35,258-321): tab-delimited text without headers, one row
per line, '%.8G' for floats, '.gz' by suffix (numpy handles it), a single row read back as shape (1, n).
Extension of this build: names ending in '.npy' are read / written as binary numpy arrays --
parsing a 20k x 100k TSV takes minutes, the association itself milliseconds.
"""
import logging

import numpy as np

fmt_float = '%.8G'
fmt_int = '%i'


def _is_binary(name):
	return name.endswith('.npy')


def file_read_tsv(f, delimiter='\t', **ka):
	"""Matrix from a TSV (or .npy) file, always 2-D."""
	logging.debug('Start reading file ' + f)
	if _is_binary(f):
		ans = np.load(f, allow_pickle=False)
		if 'dtype' in ka:
			ans = ans.astype(ka['dtype'], copy=False)
	else:
		ans = np.loadtxt(f, delimiter=delimiter, **ka)
	logging.debug('Finish reading file ' + f)
	return ans.reshape(1, -1) if ans.ndim < 2 else ans


def file_write_tsv(f, d, delimiter='\t', fmt=fmt_float, **ka):
	"""Matrix or vector to a TSV (or .npy) file."""
	logging.debug('Start writing file ' + f)
	if _is_binary(f):
		np.save(f, np.asarray(d), allow_pickle=False)
	else:
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
