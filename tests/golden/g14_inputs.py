"""Inputs of golden G14, rebuilt from a seed on either side: BASELINE configs[1] at full size -- 5000 genes x 10 000 cells in fp32 with
a shared latent factor, covariates two N(0,1) rows and an intercept.  The reference is given the SAME fp32 values
upcast to fp64 (its own fp32 arithmetic is not accurate enough to be a reference); G14 keeps its outputs for 12 of these gene rows."""
import numpy as np


def g14_inputs(seed=14, ng=5000, n=10000):
	rng = np.random.default_rng(seed)
	dt = rng.standard_normal((ng, n), dtype=np.float32)
	dt += np.float32(0.3) * rng.standard_normal((ng, 1), dtype=np.float32) * rng.standard_normal((1, n), dtype=np.float32)
	dc = np.vstack([rng.standard_normal((2, n)), np.ones((1, n))])
	rows = np.sort(rng.choice(ng, 24, replace=False))
	return dt, dc, rows
