"""Cost of contracting a block pair in S cell chunks (nrm_gram_i8_chunk, accumulating) against one launch over all cells.
Usage: time_chunks.py [rows cells [chunks ...]]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd.engine import get_engine
eng = get_engine()
rows, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1792, 10000)
Ss = [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8]
g = torch.Generator(device='cuda').manual_seed(5)
x = torch.randn((rows, n), dtype=torch.float32, device='cuda', generator=g)
y = torch.randn((rows, n), dtype=torch.float32, device='cuda', generator=g)
dc = np.ones((1, n))
from normalisr_amd.association import _prepare_covariates
dc64, dci, dcr = _prepare_covariates(dc)
d_c, d_dci = eng.covariates(dc64, dci)
rp = (rows + 127) // 128 * 128


def timeit(f, reps=20):
	for _ in range(3):
		f()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(reps):
		f()
	e1.record()
	torch.cuda.synchronize()
	return e0.elapsed_time(e1) / reps


a = eng.residualize(x, d_c, d_dci, dcr, rows_pad=rp, nslices=6, keep_fp64=False)
b = eng.residualize(y, d_c, d_dci, dcr, rows_pad=rp, nslices=6, keep_fp64=False)
dot = torch.empty((rp, rp), dtype=torch.float64, device='cuda')
ref = eng.gram(a, b, False, dot=dot, nslices=6).clone()
t1 = timeit(lambda: eng.gram(a, b, False, dot=dot, nslices=6))
tk = timeit(lambda: eng.residualize(x, d_c, d_dci, dcr, rows_pad=rp, nslices=6, keep_fp64=False))
print('%d x %d rows, %d cells: one launch %.3f ms; K1 %.3f ms' % (rows, rows, n, t1, tk))
for S in Ss:
	ac = eng.residualize_chunked(x, d_c, d_dci, dcr, rp, 6, S)
	bc = eng.residualize_chunked(y, d_c, d_dci, dcr, rp, 6, S)
	nch = len(ac._quant[0])

	def run():
		for c in range(nch):
			eng.gram_chunk(ac, bc, False, c, dot, c > 0)
	run()
	err = ((dot[:rows, :rows] - ref[:rows, :rows]).abs().max() / ref[:rows, :rows].abs().max()).item()
	tkc = timeit(lambda: eng.residualize_chunked(x, d_c, d_dci, dcr, rp, 6, S))
	print('  %d chunks of %d k-steps: %.3f ms (%.2fx), K1 chunked %.3f ms, max rel diff %.1e' % (nch, ac.cks, timeit(run), timeit(run) / t1, tkc, err))

# all K full partner blocks of a rank in one launch per chunk (gram_chunk_blocks), as CoexPlan does from 5 ranks on
K = 3
print('merged: %d x (%d x %d) block pairs per launch' % (K, rows, rows))
for S in Ss:
	ac = eng.residualize_chunked(x, d_c, d_dci, dcr, rp, 6, S)
	nch = len(ac._quant[0])
	world = K + 1
	g_chunks = [torch.stack([ac._quant[0][c]] * world) for c in range(nch)]
	g_exps = torch.stack([ac._quant[1]] * world)
	md = torch.empty((rp, K * rp), dtype=torch.float64, device='cuda')

	def runm():
		for c in range(nch):
			eng.gram_chunk_blocks(ac, g_chunks[c], g_exps, 1, K, c, md, c > 0)
	t = timeit(runm)
	print('  %d chunks: %.3f ms for %d pairs = %.3f ms per pair (%.2fx one launch per pair)' % (nch, t, K, t / K, t / K / t1))
