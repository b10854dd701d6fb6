"""Time nrm_gram_f64 alone (symmetric, C2 shape by default) from a given build of the library.
Usage: k2_time.py [lib.so [genes cells]]  -- used for kernel experiments (tools/exp/*.so built with -DGRAM_EXP=n)."""
import sys
import torch
sys.path.insert(0, '.')
from normalisr_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != '-':
	_lib.LIB_PATH = sys.argv[1]
lib = _lib.load()
ng, n = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5000, 10000)
mp, kp = (ng + 127) // 128 * 128, (n + 15) // 16 * 16
a = torch.zeros((mp, kp), dtype=torch.float64, device='cuda')
a[:ng, :n] = torch.randn((ng, n), dtype=torch.float64, device='cuda')
dot = torch.empty((mp, mp), dtype=torch.float64, device='cuda')
work = torch.empty(int(lib.nrm_gram_workspace_bytes()) // 8, dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def run():
	_lib.check(lib.nrm_gram_f64(a.data_ptr(), a.data_ptr(), mp, mp, kp, kp, kp, dot.data_ptr(), mp, 1, ng, ng, work.data_ptr(), st))
for _ in range(5):
	run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
	run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 30
print('%s: %.3f ms  %.1f TF algorithmic (%.1f executed)' % (sys.argv[1] if len(sys.argv) > 1 else 'default', ms, ng * (ng + 1) * n / ms / 1e9,
															  (mp // 128) * (mp // 128 + 1) / 2 * 128 * 128 * 2 * kp / ms / 1e9))
