"""Command runners and matrix IO for `normalisr de` / `normalisr coex` / `normalisr binnet` (mirror of the
reference's run module, run.py:20-35,258-321).  Default format as in the reference: tab-delimited text, no headers,
one row per line; outputs use '%.8G'; a '.gz' suffix selects gzip (numpy handles it).  Extension of this build
(SURVEY 8f-4): file names ending in '.npy' are read/written as binary numpy arrays -- parsing a 20k x 100k TSV
takes minutes, the association itself milliseconds."""
import logging

import numpy as np

fmt_float = '%.8G'
fmt_int = '%i'


def file_read_tsv(f, delimiter='\t', **ka):
	"""Load a TSV matrix; a single row comes back as shape (1, n) (run.py:20-27)."""
	logging.debug('Start reading file ' + f)
	if f.endswith('.npy'):
		ans = np.load(f, allow_pickle=False)
		if 'dtype' in ka:
			ans = ans.astype(ka['dtype'], copy=False)
	else:
		ans = np.loadtxt(f, delimiter=delimiter, **ka)
	logging.debug('Finish reading file ' + f)
	if ans.ndim == 1:
		ans = ans.reshape(1, -1)
	elif ans.ndim == 0:
		ans = ans.reshape(1, 1)
	return ans


def file_write_tsv(f, d, delimiter='\t', fmt=fmt_float, **ka):
	"""Write a matrix or vector as TSV with '%.8G' (run.py:30-35)."""
	logging.debug('Start writing file ' + f)
	if f.endswith('.npy'):
		ans = np.save(f, np.asarray(d), allow_pickle=False)
	else:
		ans = np.savetxt(f, d, delimiter=delimiter, fmt=fmt, **ka)
	logging.debug('Finish writing file ' + f)
	return ans


def _common_kwargs(args):
	ka = dict()
	if args.get('nth') is not None:
		ka['nth'] = args['nth']
	if args.get('bs') is not None:
		ka['bs'] = args['bs']
	if args.get('dimr') is not None:
		ka['dimreduce'] = args['dimr']
	return ka


def de(args):
	from .de import de as de_func
	dg = file_read_tsv(args['design_in'])
	dt = file_read_tsv(args['exp_in'])
	dc = file_read_tsv(args['cov_in'])
	ka = _common_kwargs(args)
	if args.get('method') is not None:
		try:
			ka['single'] = {'ignore': 0, 'single': 1, 'covariate': 4}[args['method']]
		except KeyError:
			raise ValueError('Unknown method {}'.format(args['method']))
	if args.get('clfc_out') is not None:
		ka['lowmem'] = False  # the reference leaves lowmem=True here and crashes writing None
	logging.debug('Start calculation.')
	ans = de_func(dg, dt, dc, **ka)
	logging.debug('Finish calculation.')
	file_write_tsv(args['pv_out'], ans[0])
	file_write_tsv(args['lfc_out'], ans[1])
	if args.get('clfc_out') is not None:
		file_write_tsv(args['clfc_out'], ans[2].reshape(ans[2].shape[0], -1))  # row-major (gene, covariate) per predictor
	if args.get('vard_out') is not None:
		file_write_tsv(args['vard_out'], ans[3])
	if args.get('vart_out') is not None:
		file_write_tsv(args['vart_out'], ans[4])


def coex(args):
	from .coex import coex as coex_func
	dt = file_read_tsv(args['exp_in'])
	dc = file_read_tsv(args['cov_in'])
	ka = _common_kwargs(args)
	logging.debug('Start calculation.')
	ans = coex_func(dt, dc, **ka)
	logging.debug('Finish calculation.')
	file_write_tsv(args['pv_out'], ans[0])
	if args.get('dot_out') is not None:
		file_write_tsv(args['dot_out'], ans[1])
	if args.get('var_out') is not None:
		file_write_tsv(args['var_out'], ans[2])


def normvar(args):
	from .norm import normvar as normvar_func
	dt = file_read_tsv(args['lcpm_in'])
	dc = file_read_tsv(args['cov_in'])
	dmult = file_read_tsv(args['weights_in']).ravel()
	dw = file_read_tsv(args['scale_in']).ravel()
	ka = dict()
	if args.get('nth') is not None:
		ka['nth'] = args['nth']
	if args.get('bs') is not None:
		ka['bs'] = args['bs']
	logging.debug('Start calculation.')
	ans = normvar_func(dt, dc, dmult, dw, **ka)
	logging.debug('Finish calculation.')
	file_write_tsv(args['exp_out'], ans[0])
	file_write_tsv(args['cov_out'], ans[1])


def binnet(args):
	from .binnet import binnet as binnet_func
	net = file_read_tsv(args['pv_in'])
	logging.debug('Start calculation.')
	ans = binnet_func(net, args['qcut'])
	logging.debug('Finish calculation.')
	file_write_tsv(args['net_out'], ans.astype('u1', copy=False), fmt=fmt_int)


assert __name__ != "__main__"
