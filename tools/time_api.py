"""Where the time of one numpy -> numpy coex call goes (upload, K1, K2, K3, download).  Usage: time_api.py [genes cells]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
import normalisr_amd.normalisr as norm
from normalisr_amd.engine import get_engine
from normalisr_amd.association import inv_rank

ng, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20000, 8000)
rng = np.random.default_rng(0)
dt = rng.standard_normal((ng, n), dtype=np.float32)
dc = np.vstack([rng.standard_normal((2, n)), np.ones((1, n))])
eng = get_engine()
sync = torch.cuda.synchronize


def T(f):
	sync(); t0 = time.perf_counter(); r = f(); sync(); return r, (time.perf_counter() - t0) * 1e3


for rep in range(3):
	_, t_all = T(lambda: norm.coex(dt, dc))
	dci, rank = inv_rank(dc @ dc.T)
	cov = eng.covariates(dc, dci)
	x, t_up = T(lambda: eng.upload(dt))
	ns = eng.gram_slices(n)  # 6: the integer engine (from 2048 cells on)
	rx, t_k1 = T(lambda: eng.residualize(x, cov[0], cov[1], rank, nslices=ns, keep_fp64=not ns))
	dot, t_k2 = T(lambda: eng.gram(rx, rx, True, nslices=ns))
	sw, t_k3 = T(lambda: eng.sweep(dot, rx.ss, rx.ss, ng, ng, n, n - 1 - rank, True, 0, np.float32, fix=eng.fix_args(rx, rx)))
	p, t_dp = T(lambda: eng.download(sw[0]))
	s, t_ds = T(lambda: eng.download(sw[1]))
	print(f'{ng}x{n}: coex() {t_all:.1f} ms | upload {t_up:.1f}  K1 {t_k1:.1f}  K2 {t_k2:.1f}  K3 {t_k3:.1f}  D2H p {t_dp:.1f}  D2H dot {t_ds:.1f}', flush=True)
	del x, rx, dot, sw, p, s
