import sys, numpy as np, torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
lib = _lib.load()
ng, n = 256, 10000
mp, kp = 256, 10000
g = torch.Generator(device='cuda').manual_seed(1)
a = torch.zeros((mp, kp), dtype=torch.float64, device='cuda')
a[:ng, :n] = torch.randn((ng, n), dtype=torch.float64, device='cuda', generator=g) * torch.exp(torch.randn((ng, 1), dtype=torch.float64, device='cuda', generator=g))
a[:ng, :n] += 0.3 * torch.randn((ng, 1), dtype=torch.float64, device='cuda', generator=g) * torch.randn((1, n), dtype=torch.float64, device='cuda', generator=g)
work = torch.empty(int(lib.nrm_gram_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
ref = torch.empty((mp, mp), dtype=torch.float64, device='cuda')
_lib.check(lib.nrm_gram_f64(a.data_ptr(), a.data_ptr(), mp, mp, kp, kp, kp, ref.data_ptr(), mp, 0, ng, ng, work.data_ptr(), st))
h = a.cpu().numpy().astype(np.longdouble)
exact = h @ h.T
nrm = np.sqrt(np.diag(exact))
den = np.outer(nrm, nrm)
print('fp64 kernel vs 80-bit: %.2e' % float(np.max(np.abs(ref.cpu().numpy() - exact) / den)))
for ns in (6, 5):
	q = torch.empty(int(lib.nrm_quant_bytes(mp, kp, ns)), dtype=torch.uint8, device='cuda')
	ex = torch.empty(mp, dtype=torch.int32, device='cuda')
	dot = torch.empty((mp, mp), dtype=torch.float64, device='cuda')
	_lib.check(lib.nrm_quantize_rows(a.data_ptr(), mp, kp, kp, ns, q.data_ptr(), ex.data_ptr(), 0, 0, st))
	_lib.check(lib.nrm_gram_i8_band(q.data_ptr(), ex.data_ptr(), 0, q.data_ptr(), ex.data_ptr(), 0, mp, mp, kp, ns, dot.data_ptr(), mp, 0, ng, ng, 0, mp, work.data_ptr(), st))
	d = dot.cpu().numpy()
	e = np.abs(d - exact) / den
	print('i8 x%d vs 80-bit: max %.2e  median %.2e' % (ns, float(e.max()), float(np.median(e))))
	# quantised operands reconstructed on the host: is the kernel exact for them?
	B = 8 * ns - 2
	exh = ex.cpu().numpy().astype(np.int64)
	qh = np.rint(np.ldexp(a.cpu().numpy(), -exh[:, None])).astype(np.longdouble)
	ex2 = (qh @ qh.T) * np.ldexp(np.ones(1, dtype=np.longdouble), 0)
	sc = np.ldexp(np.ones((mp, mp)), (exh[:, None] + exh[None, :])).astype(np.longdouble)
	e2 = np.abs(d - ex2 * sc) / den
	print('   vs exact product of the quantised rows: max %.2e' % float(e2.max()))
