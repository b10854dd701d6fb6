"""Time nrm_assoc_sweep alone (symmetric, C2 shape, fp32 outputs) from a given build: without row records (fp64 Gram kernels), with
the integer engine's correction only (guard tolerance 0), and with correction + guard.  Usage: k3_time.py [lib.so|-] [genes] [common factor weight]"""
import sys
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != '-':
	_lib.LIB_PATH = sys.argv[1]
lib = _lib.load()
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
n = 10000
mp = (ng + 255) // 256 * 256
g = torch.Generator(device='cuda').manual_seed(3)
x = torch.randn((ng, n), dtype=torch.float64, device='cuda', generator=g)
if len(sys.argv) > 3:  # a common factor of this weight in every row: R^2 = (w^2 / (1 + w^2))^2 for every pair
	x += float(sys.argv[3]) * torch.randn((1, n), dtype=torch.float64, device='cuda', generator=g)
dot = torch.zeros((mp, mp), dtype=torch.float64, device='cuda')
dot[:ng, :ng] = x @ x.T
ss = torch.zeros(mp, dtype=torch.float64, device='cuda')
ss[:ng] = (x * x).sum(1)
p = torch.empty((ng, ng), dtype=torch.float32, device='cuda')
st = torch.empty((ng, ng), dtype=torch.float32, device='cuda')
flags = torch.zeros(4, dtype=torch.int32, device='cuda')
fix = torch.zeros((mp, 8), dtype=torch.float64, device='cuda')
fix[:, :5] = torch.randn((mp, 5), dtype=torch.float64, device='cuda', generator=g) * 1e-9
fix[:, 5] = 8e-12
fix[:, 6] = 6e-14
s = torch.cuda.current_stream().cuda_stream
for what, ns, tol in (('no records', 0, 0.0), ('correction', 6, 0.0), ('correction + guard', 6, 2.5e-7)):
	def run():
		_lib.check(lib.nrm_assoc_sweep(dot.data_ptr(), mp, ss.data_ptr(), ss.data_ptr(), ng, ng, n, float(n - 4), 1, 0, p.data_ptr(), st.data_ptr(), 0, 0, 0, ng,
									   flags.data_ptr(), ns, fix.data_ptr(), fix.data_ptr(), tol, s))
	for _ in range(3):
		run()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(20):
		run()
	e1.record()
	torch.cuda.synchronize()
	print('%s, %s: %.4f ms  flags %s' % (sys.argv[1] if len(sys.argv) > 1 else 'default', what, e0.elapsed_time(e1) / 20, flags.cpu().tolist()))
