"""User API facade: `import normalisr_amd.normalisr as norm` (reference normalisr.py:3-9).
Only the linear-association hot path (de, coex) its direct consumer binnet and its direct producer normvar are provided; the reference's pre/post-processing
steps (qc_reads, lcpm, normcov, gotop, ...) are outside this build's scope."""
from .de import de
from .coex import coex
from .binnet import binnet
from .norm import normvar

_OUT_OF_SCOPE = ('qc_reads', 'qc_outlier', 'lcpm', 'scaling_factor', 'normcov', 'compute_var', 'gotop', 'pccovt')


def __getattr__(name):
	if name in _OUT_OF_SCOPE:
		raise NotImplementedError('normalisr_amd only provides the association hot path (de, coex); '
								  '{} is not part of this build.'.format(name))
	raise AttributeError(name)


assert __name__ != "__main__"
