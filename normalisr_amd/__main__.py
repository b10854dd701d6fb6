"""`python -m normalisr_amd <cmd>` / `normalisr <cmd>`: command line of the association hot path.
Same sub-commands, positionals and flags as the reference CLI for `de` (__main__.py:358-436) and
`coex` (:442-492), `binnet` (:498-509) and `normvar` (:311-352), global -v (:14-17), help on stderr + exit 1 without arguments (:649-651)."""
import argparse
import logging
import sys


def build_parser():
	p0 = argparse.ArgumentParser(prog='normalisr', description='Normalisr association testing (DE, co-expression) on AMD MI355X. '
								 'Only the linear-association sub-commands de and coex are provided by this build.')
	p0.add_argument('-v', dest='verbose', action='store_true', help='Verbose mode.')
	sub = p0.add_subparsers(help='sub-commands', dest='cmd')

	p = sub.add_parser('de', help='Differential expression analysis.')
	p.add_argument('design_in', help='Design/predictor matrix (predictors x cells), TSV without row or column names.')
	p.add_argument('exp_in', help='Normalized expression matrix (genes x cells), TSV.')
	p.add_argument('cov_in', help='Covariate matrix (covariates x cells), TSV.')
	p.add_argument('pv_out', help='Output P-value matrix (predictors x genes), TSV.')
	p.add_argument('lfc_out', help='Output log fold change matrix (predictors x genes), TSV.')
	p.add_argument('-m', dest='method', action='store', default='ignore',
				   help='Treatment of the other predictors when testing one: "ignore" (default), "single" (only cells with all other '
				   'predictors == 0; low-MOI screens), "covariate" (other predictors as covariates; high-MOI screens).')
	p.add_argument('-n', dest='nth', action='store', type=int, default='0', help='Number of CPU cores (kept for compatibility; the GPU path ignores it).')
	p.add_argument('-b', dest='bs', action='store', type=int, help='Batch size (kept for compatibility; results do not depend on it).')
	p.add_argument('-d', dest='dimr', action='store', type=int, help='Extra dimension loss in the expression data due to preprocessing. Default: 0.')
	p.add_argument('--clfc_out', dest='clfc_out', action='store', help='Output covariate log fold changes, (predictors, genes*covariates) row-major, TSV.')
	p.add_argument('--vard_out', dest='vard_out', action='store', help='Output variance of each predictor unexplained by covariates, TSV.')
	p.add_argument('--vart_out', dest='vart_out', action='store', help='Output variance of expression unexplained by covariates (predictors x genes), TSV.')
	p.add_argument('--gpus', dest='gpus', action='store', type=int, default=1, help='GPUs of this node to shard the problem over (one process per GPU, RCCL); every rank reads only its gene rows of exp_in. Default: 1.')

	p = sub.add_parser('coex', help='Co-expression analysis.')
	p.add_argument('exp_in', help='Normalized expression matrix (genes x cells), TSV.')
	p.add_argument('cov_in', help='Covariate matrix (covariates x cells), TSV.')
	p.add_argument('pv_out', help='Output P-value matrix (genes x genes), TSV.')
	p.add_argument('-n', dest='nth', action='store', type=int, default='0', help='Number of CPU cores (kept for compatibility; the GPU path ignores it).')
	p.add_argument('-b', dest='bs', action='store', type=int, help='Batch size (kept for compatibility; results do not depend on it).')
	p.add_argument('-d', dest='dimr', action='store', type=int, help='Extra dimension loss in the expression data due to preprocessing. Default: 0.')
	p.add_argument('--var_out', dest='var_out', action='store', help='Output variance of each gene unexplained by covariates, TSV.')
	p.add_argument('--dot_out', dest='dot_out', action='store',
				   help='Output covariance of gene pairs after covariate removal (inner product / cell count), TSV. Pearson R = dot/sqrt(var_i var_j).')
	p.add_argument('--gpus', dest='gpus', action='store', type=int, default=1, help='GPUs of this node to shard the problem over (one process per GPU, RCCL); every rank reads only its gene rows of exp_in. Default: 1.')
	p = sub.add_parser('normvar', help='Normalize variances of gene expressions and covariates.')
	p.add_argument('lcpm_in', help='Input Bayesian logCPM matrix (genes x cells), TSV.')
	p.add_argument('weights_in', help='Input vector of the fitted weight of each cell, TSV.')
	p.add_argument('cov_in', help='Input covariate matrix (covariates x cells), TSV.')
	p.add_argument('scale_in', help='Input vector of the variance-normalisation scaling factor of each gene, TSV.')
	p.add_argument('exp_out', help='Output normalized expression matrix, same format as lcpm_in.')
	p.add_argument('cov_out', help='Output normalized covariate matrix.')
	p.add_argument('-n', dest='nth', action='store', type=int, default='0', help='Number of CPU cores (kept for compatibility; ignored).')
	p.add_argument('-b', dest='bs', action='store', type=int, help='Batch size (kept for compatibility; ignored).')

	p = sub.add_parser('binnet', help='Binarize P-value co-expression network.')
	p.add_argument('pv_in', help='Input P-value matrix of gene pairwise co-expression (genes x genes), TSV.')
	p.add_argument('net_out', help='Output binary co-expression network (genes x genes, 0/1), TSV.')
	p.add_argument('qcut', type=float, help='Q-value cutoff for binary network.')
	return p0


def main(argv=None):
	argv = sys.argv[1:] if argv is None else argv
	p0 = build_parser()
	if len(argv) == 0:
		p0.print_help(sys.stderr)
		return 1
	args = vars(p0.parse_args(argv))
	logging.basicConfig(format='%(levelname)s:%(process)d:%(asctime)s:%(pathname)s:%(lineno)d:%(message)s',
						level=logging.DEBUG if args['verbose'] else logging.WARNING)
	if args['cmd'] is None:
		p0.print_help(sys.stderr)
		return 1
	if args.get('gpus', 1) < 1:
		raise ValueError('--gpus must be positive')
	if args.get('gpus', 1) > 1:  # one process per GPU, started before anything in this process touches a GPU
		from . import launch
		return launch.run_sharded(args['cmd'], args)
	from . import run
	getattr(run, args['cmd'])(args)
	return 0


if __name__ == '__main__':
	sys.exit(main())
