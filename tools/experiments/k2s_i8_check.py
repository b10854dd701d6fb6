"""Kernel-level check and timing of the streaming pass on the int8 matrix cores (nrm_skinny_i8) against the fp64 one (nrm_gram_skinny).
Usage: k2s_i8_check.py [rows cells nz dtype(f32|f64)]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
lib = _lib.load()
rows, n, nz = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20000, 100000, 21)
dt = torch.float64 if (len(sys.argv) > 4 and sys.argv[4] == 'f64') else torch.float32
code = 1 if dt == torch.float64 else 0
g = torch.Generator(device='cuda').manual_seed(7)
y = (torch.randn((rows, n), dtype=dt, device='cuda', generator=g) * 0.7 + 2.0)
k32 = (n + 127) // 128 * 128
z = torch.zeros((32, k32), dtype=torch.float64, device='cuda')
z[:nz, :n] = torch.randn((nz, n), dtype=torch.float64, device='cuda', generator=g)
z[nz - 2, :n] = 1.0
z[31] = 1.0
rp = (rows + 255) // 256 * 256
st = torch.cuda.current_stream().cuda_stream
G0 = torch.zeros((rp, 32), dtype=torch.float64, device='cuda')
ss0 = torch.zeros(rp, dtype=torch.float64, device='cuda')
w0 = torch.empty(int(lib.nrm_gram_skinny_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
f64 = lambda: _lib.check(lib.nrm_gram_skinny(y.data_ptr(), code, rows, n, y.stride(0), z.data_ptr(), k32, k32, G0.data_ptr(), ss0.data_ptr(), rp, nz, 0.0, w0.data_ptr(), st))
ysh = torch.empty(rows, dtype=torch.int32, device='cuda')
ssr = torch.empty(rows, dtype=torch.float64, device='cuda')
scales = lambda: _lib.check(lib.nrm_row_scales(y.data_ptr(), code, rows, n, y.stride(0), ysh.data_ptr(), ssr.data_ptr(), st))
nks = k32 // 32
planes = torch.empty(6 * nks * 1024, dtype=torch.uint8, device='cuda')
zsh = torch.empty(32, dtype=torch.int32, device='cuda')
zfix = torch.empty((32, 8), dtype=torch.float64, device='cuda')
quant = lambda: _lib.check(lib.nrm_quantize_rows(z.data_ptr(), 32, k32, k32, 6, planes.data_ptr(), zsh.data_ptr(), zfix.data_ptr(), n, st))
G1 = torch.zeros((rp, 32), dtype=torch.float64, device='cuda')
ss1 = torch.zeros(rp, dtype=torch.float64, device='cuda')
dig = torch.zeros((rp, 8), dtype=torch.float64, device='cuda')
w1 = torch.empty(int(lib.nrm_skinny_i8_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
i8 = lambda: _lib.check(lib.nrm_skinny_i8(y.data_ptr(), code, rows, n, y.stride(0), ysh.data_ptr(), planes.data_ptr(), zsh.data_ptr(), k32, G1.data_ptr(), ss1.data_ptr(),
										  dig.data_ptr(), rp, w1.data_ptr(), st))
f64(); scales(); quant(); i8()
torch.cuda.synchronize()
nrm = torch.sqrt(ss0[:rows])[:, None] * torch.sqrt((z[:nz] ** 2).sum(1))[None, :]
# mean-product correction on the host, as the sweep applies it
sh = ysh.double()
uy = torch.stack([torch.ldexp(dig[:rows, s], (ysh + 8 * s)) for s in range(5)], 1)  # (rows, 5)
vz = torch.cumsum(zfix[:nz, :5], 1)  # prefix sums (nz, 5)
corr = sum(uy[:, s:s + 1] * vz[:, 4 - s][None, :] for s in range(5)) / n
err_raw = ((G1[:rows, :nz] - G0[:rows, :nz]).abs() / nrm).max().item()
err_fix = ((G1[:rows, :nz] + corr - G0[:rows, :nz]).abs() / nrm).max().item()
print('max |G_i8 - G_f64| / (|y||z|): raw %.2e, with the mean-product correction %.2e;  ss rel diff %.2e, vs row_scales %.2e' % (
	err_raw, err_fix, ((ss1[:rows] - ss0[:rows]).abs() / ss0[:rows]).max().item(), ((ssr - ss0[:rows]).abs() / ss0[:rows]).max().item()))
# digit sums against a host quantisation of a few rows
for r in (0, rows // 2, rows - 1):
	q = torch.round(torch.ldexp(y[r].double(), -ysh[r])).cpu().numpy().astype(np.int64)
	want = []
	for s in range(5):
		d = ((q & 0xff) ^ 0x80) - 0x80
		q = (q - d) >> 8
		want.append(int(d.sum()))
	print('row', r, 'digit sums', [int(v) for v in dig[r, :5].cpu().numpy()], 'host', want)
def timeit(f, reps=10):
	for _ in range(2):
		f()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(reps):
		f()
	e1.record()
	torch.cuda.synchronize()
	return e0.elapsed_time(e1) / reps
gb = rows * n * y.element_size() / 1e9
for name, f in (('fp64 kernel', f64), ('int8 kernel', i8), ('row scales', scales), ('quantise Z', quant)):
	ms = timeit(f)
	print('%-12s %.3f ms  (%.2f TB/s of expression data)' % (name, ms, gb / ms))
