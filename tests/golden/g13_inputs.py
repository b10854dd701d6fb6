"""Inputs of golden G13, rebuilt from a seed on either side (the reference run in the build container by make_golden.py, the device
run on the GPU box by tests/test_gpu_round3.py): 100 000 cells are too many to store, the reference's outputs for 40 genes are not.
log1p(Poisson) counts whose rate follows two latent factors and a batch, from 1 % dense to dense; covariates: three one-hot batches
of four (the fourth is the intercept's), log total counts, an intercept; three groupings (3 %, 30 %, one batch-correlated)."""
import numpy as np


def g13_inputs(seed=13, n=100000, ng=40):
	rng = np.random.default_rng(seed)
	lat = rng.normal(size=(2, n))
	batch = rng.integers(0, 4, n)
	lam = np.geomspace(0.01, 8, ng)[:, None] * np.exp(0.5 * rng.normal(size=(ng, 2)) @ lat + 0.2 * rng.normal(size=(ng, 4))[:, batch] - 0.15)
	counts = rng.poisson(lam)
	dt = np.log1p(counts.astype(np.float64))
	tot = np.log(counts.sum(axis=0) + 1.0)
	dc = np.vstack([(batch[None, :] == np.arange(3)[:, None]).astype(np.float64), (tot - tot.mean()) / tot.std(), np.ones(n)])
	dg = np.vstack([(rng.random(n) < 0.03), (rng.random(n) < 0.3), (rng.random(n) < 0.05 + 0.1 * (batch == 2))]).astype(np.float64)
	return dt, dc, dg
