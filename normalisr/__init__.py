"""Drop-in import name: `import normalisr.normalisr as norm` resolves to the MI355X build
(normalisr_amd) for the association hot path.  Put this repository root on PYTHONPATH instead of the
reference's src/ directory."""
import importlib
import sys

__all__ = ['association', 'binnet', 'coex', 'de', 'norm', 'normalisr', 'parallel', 'run']
for _m in __all__:
	sys.modules[__name__ + '.' + _m] = importlib.import_module('normalisr_amd.' + _m)
	globals()[_m] = sys.modules[__name__ + '.' + _m]
del _m
