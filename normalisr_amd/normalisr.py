"""User API facade: `import normalisr_amd.normalisr as norm` (reference normalisr.py:3-9).
Only the linear-association hot path (de, coex) and its direct consumer binnet are provided; the reference's pre/post-processing
steps (qc_reads, lcpm, normcov, normvar, gotop, ...) are outside this build's scope."""
from .de import de
from .coex import coex
from .binnet import binnet

_OUT_OF_SCOPE = ('qc_reads', 'qc_outlier', 'lcpm', 'scaling_factor', 'normcov', 'compute_var', 'normvar', 'gotop', 'pccovt')


def __getattr__(name):
	if name in _OUT_OF_SCOPE:
		raise NotImplementedError('normalisr_amd only provides the association hot path (de, coex); '
								  '{} is not part of this build.'.format(name))
	raise AttributeError(name)


assert __name__ != "__main__"
