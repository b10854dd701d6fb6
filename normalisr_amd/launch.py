"""`normalisr coex|de ... --gpus N`: the sharded form of the CLI (reference: run.py:289-310, __main__.py:442-492 on one host).

The parent process never touches a GPU: it starts N rank processes (one per GPU, `python -m normalisr_amd.shard_worker`) with the
torch.distributed environment of a one-node job on 127.0.0.1, waits for them, and turns any failing rank into a non-zero exit
status (the survivors are stopped: a rank waiting in a collective for a dead peer would wait for ever).  Every rank reads ONLY its
block of gene rows of the expression matrix (memory-mapped .npy, or the matching lines of a TSV), the ranks exchange what the
pair space needs over RCCL (normalisr_amd.distributed), each copies its rows of the results into arrays shared by the ranks, and
rank 0 writes the output files in the requested format.

Environment: NRM_DIST_BACKEND (nccl = RCCL, default; gloo for functional runs), NRM_SHARE_GPU=1 (all ranks on GPU 0: functional
runs on a one-GPU box).
"""
import json
import os
import socket
import subprocess
import sys
import time

from . import _opts

SHARDED = ('coex', 'de')


def check_args(cmd, args):
	if cmd not in SHARDED:
		raise ValueError('--gpus applies to the sub-commands {}'.format(', '.join(SHARDED)))
	if cmd == 'de' and args.get('method', 'ignore') not in ('ignore', 'single', 'covariate'):
		raise ValueError('Unknown method {}'.format(args['method']))
	if cmd == 'de' and args.get('clfc_out') is not None:
		raise ValueError('--clfc_out is not provided with --gpus > 1')


def run_sharded(cmd, args):
	"""Start args['gpus'] ranks of sub-command `cmd`; returns the exit status (0 only when every rank succeeded)."""
	n = int(args['gpus'])
	check_args(cmd, args)
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	payload = json.dumps(dict(cmd=cmd, args=args))
	procs = []
	for r in range(n):
		env = dict(os.environ)
		env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
		env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
		env['PYTHONPATH'] = root + (os.pathsep + env['PYTHONPATH'] if env.get('PYTHONPATH') else '')
		procs.append(subprocess.Popen([sys.executable, '-m', 'normalisr_amd.shard_worker', payload], env=env))
	status = 0
	live = list(procs)
	grace = float(_opts.debug('rank_grace_s', '20'))  # what the survivors get to exit by themselves after the first failure
	limit = float(_opts.debug('job_timeout_s', '0'))  # overall watchdog, 0 = none
	t0 = time.time()
	failed_at = None
	while live:
		time.sleep(0.05)
		for p in list(live):
			rc = p.poll()
			if rc is None:
				continue
			live.remove(p)
			if rc != 0 and status == 0:
				status = rc if rc > 0 else 1
				failed_at = time.time()
				for q in live:  # a dead peer leaves the others waiting in their next collective
					q.terminate()
		now = time.time()
		if limit > 0 and now - t0 > limit and status == 0:
			status, failed_at = 124, now
			for q in live:
				q.terminate()
		if failed_at is not None and now - failed_at > grace:
			for q in live:  # SIGTERM was not enough (a rank blocked inside a collective): kill
				q.kill()
	return status
