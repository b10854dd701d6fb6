import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
	config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
	def load(name):
		return np.load(os.path.join(GOLDEN, name + '.npz'))
	return load


def relerr(a, b, atol=0.):
	"""max |a-b| / (|b| + atol-scaled floor), elementwise relative error with absolute floor."""
	a = np.asarray(a, dtype=np.float64)
	b = np.asarray(b, dtype=np.float64)
	return float(np.max(np.abs(a - b) / (np.abs(b) + atol))) if a.size else 0.
