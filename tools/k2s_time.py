"""Time nrm_gram_skinny alone (C3 shape: 20000 rows x 100000 cells fp32 against 21 Z rows) from a given build.
Usage: k2s_time.py [lib.so [nz [constant_row_value]]]"""
import sys
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != '-':
	_lib.LIB_PATH = sys.argv[1]
lib = _lib.load()
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 21
cval = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0  # != 0: one more (constant) Z row summed on the vector ALU
ny, n = 20000, 100000
k32 = (n + 127) // 128 * 128
y = torch.zeros((ny, k32), dtype=torch.float32, device='cuda')
y[:, :n] = torch.randn((ny, n), dtype=torch.float32, device='cuda')
z = torch.zeros((32, k32), dtype=torch.float64, device='cuda')
z[:nz, :n] = torch.randn((nz, n), dtype=torch.float64, device='cuda')
ny_pad = (ny + 255) // 256 * 256
g = torch.empty((ny_pad, 32), dtype=torch.float64, device='cuda')
ss = torch.empty((ny_pad, ), dtype=torch.float64, device='cuda')
work = torch.empty(int(lib.nrm_gram_skinny_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def run():
	_lib.check(lib.nrm_gram_skinny(y.data_ptr(), 0, ny, n, y.stride(0), z.data_ptr(), k32, k32, g.data_ptr(), ss.data_ptr(), ny_pad, nz, cval, work.data_ptr(), st))
for _ in range(3):
	run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
	run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print('%s nz=%d: %.3f ms  %.2f TB/s  %.1f TF executed' % (sys.argv[1] if len(sys.argv) > 1 else 'default', nz, ms, 4.0 * ny * n / ms / 1e9, 2.0 * ny * n * (32 if nz > 16 else 16) / ms / 1e9))
