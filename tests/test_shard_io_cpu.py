"""CPU tests of the sharded CLI's host logic (normalisr_amd/shard_worker.py, launch.py): which gene rows a rank reads.
The reference reads whole files with np.loadtxt (run.py:20-27); a rank must see exactly its rows of what loadtxt would return."""
import gzip

import numpy as np
import pytest

from normalisr_amd import shard_worker as sw


@pytest.mark.parametrize('gz', [False, True])
def test_tsv_blocks_equal_loadtxt_rows(tmp_path, gz):
	"""Comment lines, blank lines and trailing comments are dropped by loadtxt and count neither as rows nor towards a block."""
	rng = np.random.default_rng(0)
	a = rng.normal(size=(11, 5))
	lines = ['# header line', '']
	for i, row in enumerate(a):
		lines.append('\t'.join('%.17g' % v for v in row) + (' # gene %d' % i if i % 4 == 0 else ''))
		if i in (2, 7):
			lines += ['', '#' + ' x' * i]
	path = str(tmp_path / ('e.tsv.gz' if gz else 'e.tsv'))
	with (gzip.open if gz else open)(path, 'wt') as f:
		f.write('\n'.join(lines) + '\n')
	whole = np.loadtxt(path, delimiter='\t', ndmin=2)
	assert np.array_equal(whole, a)
	assert sw.matrix_rows(path) == 11
	for world in (1, 2, 3, 4, 8):
		for balanced in (False, True):
			got = [sw.read_rows(path, *sw.block_bounds(11, world, r, balanced)) for r in range(world)]
			got = [g for g in got if g.shape[0]]
			assert np.array_equal(np.vstack(got), a), (world, balanced)


def test_block_bounds_cover_every_row_once():
	for rows in (1, 8, 9, 17, 1000, 30000):
		for world in (1, 2, 3, 8):
			if rows < world:
				continue
			for balanced in (False, True):
				b = [sw.block_bounds(rows, world, r, balanced) for r in range(world)]
				assert b[0][0] == 0 and b[-1][1] == rows and all(x[1] == y[0] for x, y in zip(b[:-1], b[1:]))
				if balanced:  # de: no rank without rows
					assert all(hi > lo for lo, hi in b)
				else:  # coex: every block at most ceil(rows / world) rows, the tail ones possibly empty (padded by the worker)
					assert all(0 <= hi - lo <= -(-rows // world) for lo, hi in b)


def test_empty_block_reads_as_no_rows(tmp_path):
	path = str(tmp_path / 'e.tsv')
	np.savetxt(path, np.arange(12.).reshape(3, 4), delimiter='\t')
	assert sw.read_rows(path, 3, 3).shape[0] == 0
	np.save(str(tmp_path / 'e.npy'), np.arange(12.).reshape(3, 4))
	assert sw.read_rows(str(tmp_path / 'e.npy'), 3, 3).shape == (0, 4)
