"""Round 6, verdict item 6(a): what the ragged edge of configs[1] costs K2 and what a split launch would buy, measured with the entries the library has.
5000 genes = 39 x 128 + 8: the 40 tiles of the last tile column hold 8 valid columns each; in them 2 of the 8 waves compute (both on one SIMD) and the
tile takes as long as a full one -- 40 of 820 tiles, 4.9 % of the kernel.
  (a) the shipped form: one symmetric launch over 5120 padded rows;
  (b) the symmetric launch over the first 4992 rows (780 whole tiles = 3 x 256 + 12) plus the edge as a launch of its own, TRANSPOSED: A = the last 128-row
      block (8 valid rows), B = all rows -- 40 tiles in which 4 waves (one per SIMD) compute one 32-row half each;
  (c) (b)'s first launch alone: the bound of ANY edge treatment (the edge for free).
Usage: k2_edge_exp.py [genes cells]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
lib = _lib.load()
ng, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 10000)
ns = 6
mp, kp = (ng + 127) // 128 * 128, (n + 15) // 16 * 16
g = torch.Generator(device='cuda').manual_seed(1)
a = torch.zeros((mp, kp), dtype=torch.float64, device='cuda')
a[:ng, :n] = torch.randn((ng, n), dtype=torch.float64, device='cuda', generator=g)
nks = (kp + 31) // 32
q = torch.empty(int(lib.nrm_quant_bytes(mp, kp, ns)), dtype=torch.uint8, device='cuda')
ex = torch.empty(mp, dtype=torch.int32, device='cuda')
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.nrm_quantize_rows(a.data_ptr(), mp, kp, kp, ns, q.data_ptr(), ex.data_ptr(), 0, 0, st))
work = torch.empty(int(lib.nrm_gram_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
dot = torch.empty((mp, mp), dtype=torch.float64, device='cuda')
dot2 = torch.full((mp, mp), float('nan'), dtype=torch.float64, device='cuda')
strip = torch.empty((128, mp), dtype=torch.float64, device='cuda')
plane = (mp // 32) * nks * 1024  # distance between the digit planes of the whole matrix
m0 = mp - 128
last = q[(m0 // 32) * nks * 1024:]


def whole():
	_lib.check(lib.nrm_gram_i8_band(q.data_ptr(), ex.data_ptr(), 0, q.data_ptr(), ex.data_ptr(), 0, mp, mp, kp, ns, dot.data_ptr(), mp, 1, ng, ng, 0, mp, work.data_ptr(), st))


def inner():
	_lib.check(lib.nrm_gram_i8_band(q.data_ptr(), ex.data_ptr(), plane, q.data_ptr(), ex.data_ptr(), plane, m0, m0, kp, ns, dot2.data_ptr(), mp, 1, m0, m0, 0, m0, work.data_ptr(), st))


def edge():
	_lib.check(lib.nrm_gram_i8_band(last.data_ptr(), ex[m0:].data_ptr(), plane, q.data_ptr(), ex.data_ptr(), plane, 128, mp, kp, ns, strip.data_ptr(), mp, 0, ng - m0, ng, 0, 128, work.data_ptr(), st))


def split():
	inner()
	edge()


def timeit(f, reps=30):
	for _ in range(5):
		f()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(reps):
		f()
	e1.record()
	torch.cuda.synchronize()
	return e0.elapsed_time(e1) / reps


whole()
split()
torch.cuda.synchronize()
iu = torch.triu_indices(m0, m0, device='cuda')
same_inner = bool((dot[iu[0], iu[1]] == dot2[iu[0], iu[1]]).all())
v = ng - m0
same_edge = bool((strip[:v, :ng].t() == torch.where(torch.arange(ng, device='cuda')[:, None] <= (m0 + torch.arange(v, device='cuda'))[None, :], dot[:ng, m0:ng], strip[:v, :ng].t())).all())
for rep in range(3):
	ta, tb, tc, te = timeit(whole), timeit(split), timeit(inner), timeit(edge)
	print('%d genes x %d cells: (a) one launch %.3f ms   (b) inner + transposed edge %.3f ms (edge alone %.3f)   (c) inner alone %.3f ms   inner bits equal: %s, edge bits equal: %s' % (
		ng, n, ta, tb, te, tc, same_inner, same_edge))
