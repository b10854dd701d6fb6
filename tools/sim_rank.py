"""Compute side of one rank of an N-GPU coex step, timed on ONE GPU: the exchange is replaced by a local fill of the
gathered buffer (no collective), so the numbers are the per-rank K1/K2/K3 times of bench.py --gpus N without xGMI.
Usage: sim_rank.py [world [rank]]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd.distributed import CoexPlan

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 10000
ng = int(round(5000 * np.sqrt(world) / world)) * world
R = ng // world
gen = torch.Generator(device='cuda').manual_seed(1)
full = torch.randn((ng, n), dtype=torch.float32, device='cuda', generator=gen)
dc = np.vstack([np.random.default_rng(0).standard_normal((2, n)), np.ones((1, n))])


class LocalPlan(CoexPlan):
	def _exchange(self, data, ss):
		self._blocks = {}
		self.all_x[:self.world * self.rows].copy_(full)
		return []


for merge in ('0', '1'):
	os.environ['NRM_MERGE_PARTNERS'] = merge
	plan = LocalPlan(full[rank * R:(rank + 1) * R].contiguous(), dc, rank=rank, world=world)
	for _ in range(3):
		plan.step()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(10):
		plan.step(timed=True)
	torch.cuda.synchronize()
	ms = (time.perf_counter() - t0) * 100
	pairs = plan.local_pair_count()
	print(f'world {world} rank {rank} genes {ng} rows/rank {R} merged={merge}: {ms:.2f} ms/step, {pairs / ms / 1e6:.2f}e9 pairs/s/GPU, '
		  f'gram {plan.gram_ms():.2f} ms = {2 * n * pairs / plan.gram_ms() / 1e9:.1f} TF, {plan.kernel_breakdown()}', flush=True)
