"""The command line's text matrices through the library's threaded parser / printer (csrc/nrm_tsv.hip) against numpy.loadtxt /
numpy.savetxt as the reference calls them (run.py:20-35): the same numbers in, byte for byte the same text out."""
import os

import numpy as np
import pytest

from normalisr_amd import run


def _cases():
	rng = np.random.default_rng(11)
	bits = rng.integers(0, 2**63, size=(60, 500), dtype=np.int64).view(np.float64).copy()  # every magnitude, denormals included
	bits[~np.isfinite(bits)] = 1.0
	bits[::2] *= -1
	sub = rng.integers(1, 2**52, size=(4, 500), dtype=np.int64).view(np.float64).copy() / 2.0**rng.integers(0, 50, (4, 500))  # subnormals of every width
	special = np.array([[0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 1e8, 123456789.0, 1e-5, 0.0001, 99999999.5, 5e-324, 1.7976931348623157e308,
						 1.00000005, 0.000099999999, 9.9999999e-5, 99999995, 999999995, 123456785, 1.2345678e8, 1e7, 12345678, 0.5, 2.5e-5]])
	ties = np.array([[(m * 10 + 5) * 1.0 for m in range(10000000, 10000400)]])  # 9-digit integers ending in 5: exact ties, round half to even
	return dict(bits=bits, subnormal=sub, special=special, ties=ties, ties_scaled=ties / 2**20, powers=np.array([[10.0**k for k in range(-12, 13)] + [9.99999995 * 10.0**k for k in range(-12, 13)]]),
				normal=rng.standard_normal((50, 300)), pvalues=10.0**-rng.uniform(0, 300, (50, 300)), f32=rng.standard_normal((50, 300)).astype(np.float32),
				f32_small=(10.0**-rng.uniform(0, 44, (50, 300))).astype(np.float32), vector=rng.standard_normal(37), one=np.array([[3.25]]))


@pytest.mark.parametrize('name', sorted(_cases()))
def test_text_out_is_numpy_savetxt_byte_for_byte(tmp_path, name, monkeypatch):
	x = _cases()[name]
	a, b = str(tmp_path / 'numpy.tsv'), str(tmp_path / 'ours.tsv')
	np.savetxt(a, x, delimiter='\t', fmt='%.8G')
	monkeypatch.setenv('NRM_TSV', 'native')
	run.file_write_tsv(b, x)
	assert open(a, 'rb').read() == open(b, 'rb').read()
	# ... and read back: the same numbers numpy.loadtxt reads, signs of zeros and NaNs included
	with np.errstate(all='ignore'):
		ref = np.loadtxt(a, delimiter='\t')
	got = run.file_read_tsv(a)
	ref = ref.reshape(1, -1) if ref.ndim < 2 else ref
	assert got.shape == ref.shape and got.dtype == ref.dtype
	assert np.array_equal(got, ref, equal_nan=True) and np.array_equal(np.signbit(got), np.signbit(ref))


def test_integers_gz_and_the_numpy_switch(tmp_path, monkeypatch):
	rng = np.random.default_rng(3)
	net = rng.integers(0, 2, (40, 70)).astype(bool)
	a, b = str(tmp_path / 'a.tsv'), str(tmp_path / 'b.tsv')
	for x in (net, net.astype('u1'), rng.integers(-2**40, 2**40, (9, 5)), rng.integers(-1000, 1000, (9, 5)).astype(np.int32)):
		np.savetxt(a, x, delimiter='\t', fmt='%i')
		run.file_write_tsv(b, x, fmt='%i')
		assert open(a, 'rb').read() == open(b, 'rb').read()
	x = rng.standard_normal((30, 20))
	run.file_write_tsv(b + '.gz', x)
	assert np.array_equal(run.file_read_tsv(b + '.gz'), np.loadtxt(b + '.gz', delimiter='\t'))
	monkeypatch.setenv('NRM_TSV', 'numpy')  # the reference's own calls
	run.file_write_tsv(a, x)
	monkeypatch.setenv('NRM_TSV', 'native')
	run.file_write_tsv(b, x)
	assert open(a, 'rb').read() == open(b, 'rb').read()


def test_text_in_follows_loadtxt(tmp_path):
	"""Comments, blank lines, CRLF, blanks around fields, '+', out-of-range numbers, a last line without newline; a single column comes
	back as one row (loadtxt squeezes, run.py reshapes: the reference's behaviour); malformed text raises ValueError."""
	f = str(tmp_path / 'x.tsv')
	open(f, 'w').write('# header\n1\t2\t3\r\n\n4\t5\t6 # tail\n+7\t 8 \t1e400\n-1e-400\tnan\tINF')
	got = run.file_read_tsv(f)
	with np.errstate(all='ignore'):
		ref = np.loadtxt(f, delimiter='\t')
	assert np.array_equal(got, ref, equal_nan=True) and got.shape == (4, 3) and np.signbit(got[3, 0])
	open(f, 'w').write('1\n2\n3\n')
	assert run.file_read_tsv(f).shape == (1, 3) and np.array_equal(run.file_read_tsv(f), np.loadtxt(f, delimiter='\t').reshape(1, -1))
	big = np.random.default_rng(5).standard_normal((3000, 700))  # > 1 MB of text per piece: several threads, rows dealt in order
	np.savetxt(f, big, delimiter='\t', fmt='%.17g')
	assert np.array_equal(run.file_read_tsv(f), big)
	assert np.array_equal(run.file_read_tsv(f, dtype=np.float32), big.astype(np.float32))
	for bad in ('1\t2\n3\n', '1\tx\n', '1\t2\n3\t4\t5\n', '1\t0x10\n', '1\t\t2\n'):
		open(f, 'w').write(bad)
		with pytest.raises(ValueError):
			run.file_read_tsv(f)
		with pytest.raises(ValueError):
			np.loadtxt(f, delimiter='\t')


def test_text_the_parser_does_not_take_is_numpys_to_answer(tmp_path):
	"""Round-4 advisor: strtod takes 'nan(abc)' where numpy.loadtxt raises, and a '\\r' inside a line is a line break for numpy's text reader.  The
	library's parser refuses both; run.file_read_tsv then hands the file to numpy.loadtxt, whose matrix or exception is the reference's."""
	from normalisr_amd import _lib
	f = str(tmp_path / 'x.tsv')
	for text in ('1\tnan(abc)\n', '1\t-NaN(0x7)\n', '1.5\r2\n', '1\t2\r3\t4\n', '1\t2\r\n3\t4\r\n', '1\tnan\n+NaN\t-nan\n'):
		open(f, 'w', newline='').write(text)
		try:
			with np.errstate(all='ignore'):
				ref = np.loadtxt(f, delimiter='\t')
				ref = ref.reshape(1, -1) if ref.ndim < 2 else ref  # (run.py:24-26: a single row or column comes back as one row)
		except ValueError:
			ref = None
		if ref is None:
			with pytest.raises(ValueError):
				run.file_read_tsv(f)
		else:
			assert np.array_equal(run.file_read_tsv(f), ref, equal_nan=True), text
	# the parser itself: payload NaNs and inner carriage returns are errors, plain NaNs of either sign and CRLF line ends are not
	buf = lambda t: np.frombuffer(t.encode(), dtype=np.uint8)
	for t in ('1\tnan(abc)\n', '1\r2\t3\n'):
		with pytest.raises(ValueError):
			run.parse_text(buf(t))
	got = run.parse_text(buf('1\tnan\r\n-NAN\t+nan\r\n'))
	assert got.shape == (2, 2) and got[0, 0] == 1 and np.isnan(got[0, 1]) and np.isnan(got[1]).all()
