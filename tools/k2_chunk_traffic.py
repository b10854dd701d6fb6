"""Fabric traffic of K2 over 500 000 cells in ONE launch against 8 cell-chunk launches (run under rocprofv3 --pmc FETCH_SIZE): a
3840 x 3840 block pair, one pass of each form; the launches appear in this order: 1 whole, then 8 chunks."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd.engine import get_engine
from normalisr_amd.association import _prepare_covariates
eng = get_engine()
rows, n, S = 3840, 500000, 8
g = torch.Generator(device='cuda').manual_seed(5)
x = torch.randn((rows, n), dtype=torch.float32, device='cuda', generator=g)
dc64, dci, dcr = _prepare_covariates(np.ones((1, n)))
d_c, d_dci = eng.covariates(dc64, dci)
a = eng.residualize(x, d_c, d_dci, dcr, rows_pad=rows, nslices=6, keep_fp64=False)
dot = torch.empty((rows, rows), dtype=torch.float64, device='cuda')
eng.gram(a, a, False, dot=dot, nslices=6)
torch.cuda.synchronize()
ac = eng.residualize_chunked(x, d_c, d_dci, dcr, rows, 6, S)
for c in range(len(ac._quant[0])):
	eng.gram_chunk(ac, ac, False, c, dot, c > 0)
torch.cuda.synchronize()
print('done')
