"""Time and check the integer Gram engine (quantise + nrm_gram_i8) against nrm_gram_f64 (symmetric, C2 shape by default).
Usage: k2i8_time.py [lib.so|-] [genes cells [slices]]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != '-':
	_lib.LIB_PATH = sys.argv[1]
lib = _lib.load()
ng, n = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5000, 10000)
mp, kp = (ng + 127) // 128 * 128, (n + 15) // 16 * 16
g = torch.Generator(device='cuda').manual_seed(1)
a = torch.zeros((mp, kp), dtype=torch.float64, device='cuda')
a[:ng, :n] = torch.randn((ng, n), dtype=torch.float64, device='cuda', generator=g) * torch.exp(torch.randn((ng, 1), dtype=torch.float64, device='cuda', generator=g))
a[:ng, :n] += 0.3 * torch.randn((ng, 1), dtype=torch.float64, device='cuda', generator=g) * torch.randn((1, n), dtype=torch.float64, device='cuda', generator=g)
dot = torch.empty((mp, mp), dtype=torch.float64, device='cuda')
ref = torch.empty((mp, mp), dtype=torch.float64, device='cuda')
work = torch.empty(int(lib.nrm_gram_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.nrm_gram_f64(a.data_ptr(), a.data_ptr(), mp, mp, kp, kp, kp, ref.data_ptr(), mp, 1, ng, ng, work.data_ptr(), st))
nrm = torch.sqrt((a * a).sum(dim=1))
for ns in ([int(sys.argv[4])] if len(sys.argv) > 4 else [6, 5]):
	q = torch.empty(int(lib.nrm_quant_bytes(mp, kp, ns)), dtype=torch.uint8, device='cuda')
	ex = torch.empty(mp, dtype=torch.int32, device='cuda')
	def quant():
		_lib.check(lib.nrm_quantize_rows(a.data_ptr(), mp, kp, kp, ns, q.data_ptr(), ex.data_ptr(), 0, 0, st))
	def run():
		_lib.check(lib.nrm_gram_i8_band(q.data_ptr(), ex.data_ptr(), 0, q.data_ptr(), ex.data_ptr(), 0, mp, mp, kp, ns, dot.data_ptr(), mp, 1, ng, ng, 0, mp, work.data_ptr(), st))
	dot.fill_(float('nan'))
	quant()
	run()
	torch.cuda.synchronize()
	iu = torch.triu_indices(ng, ng, device='cuda')
	d, r = dot[iu[0], iu[1]], ref[iu[0], iu[1]]
	err = ((d - r).abs() / (nrm[iu[0]] * nrm[iu[1]])).max().item()  # error of Pearson r
	def timeit(f, reps):
		for _ in range(3):
			f()
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		for _ in range(reps):
			f()
		e1.record()
		torch.cuda.synchronize()
		return e0.elapsed_time(e1) / reps
	tq, tg = timeit(quant, 10), timeit(run, 20)
	print('slices=%d: max |dot_i8 - dot_f64| / (|a_i||a_j|) = %.2e   quantise %.3f ms  gram %.3f ms  = %.1f fp64-equivalent TF algorithmic' % (
		ns, err, tq, tg, ng * (ng + 1) * n / tg / 1e9))
def f64():
	_lib.check(lib.nrm_gram_f64(a.data_ptr(), a.data_ptr(), mp, mp, kp, kp, kp, ref.data_ptr(), mp, 1, ng, ng, work.data_ptr(), st))
for _ in range(3):
	f64()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
	f64()
e1.record()
torch.cuda.synchronize()
print('fp64 kernel: %.3f ms' % (e0.elapsed_time(e1) / 10))
