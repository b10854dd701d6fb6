"""Switches of the package.

A user needs a handful, each an environment variable of its own (README.md): NRM_GRAM, NRM_I8_GUARD_TOL, NRM_DE_SPARSE, NRM_EXCHANGE,
NRM_PINNED_POOL_MB, NRM_HOST_ENTRY, NORMALISR_DEVICE, NORMALISR_BLAS_THREADS (and NRM_DIST_BACKEND / NRM_SHARE_GPU for functional runs of
the sharded paths on one GPU).  Everything else -- ablations, traces, fallbacks kept for A/B measurements -- is ONE variable,

    NRM_DEBUG="graph=0,pipeline=0,s4_start=norm"

a comma-separated list of key=value pairs (keys below, case-insensitive).  For scripts written before round 5 a key KEY is also read from
the environment variable NRM_KEY."""
import os

DEBUG_KEYS = {
	'pipeline': "0: no PCIe pipelining of numpy-in / numpy-out calls",
	'graph': "0: resident de steps are not replayed as HIP graphs",
	'const_row': "0: the intercept stays on the matrix cores of the streaming de kernel",
	'de_path': "auto | general | streaming: force / forbid the streaming de kernel",
	'de_sparse_sums': "stream: the sparse-design step takes the rows' sums with the covariates in a pass of its own (k_s1_stream) instead of inside k_de_sparse",
	'i8_fix': "0 (tests only): sweep without the integer engine's row records",
	's4_inverse': "host: single=4 inverts M~ with LAPACK instead of the on-device Newton-Schulz iteration",
	's4_start': "norm: that iteration from I / ||M~||_1 only",
	's4_sparse_m': "0: single=4 on a sparse design takes M~ from K1's residuals and the fp64 Gram kernel",
	's4_trace': "1: phase times of a single=4 call", 's1_trace': "1: phase times of a single=1 call",
	'single1': "dense: single=1 through the masked Gram contractions whatever the design",
	'single1_stats': "host: single=1 finishes the groupings' statistics (pseudo-inverses, ranks, P-value plans) on the host as rounds 3-5 did",
	's4_plan': "public: a Single4Plan calls the public function every step (no lean device-only replay)",
	'normvar': "host: normvar through the Gram launches and the host's batched pseudo-inverses",
	'small_svd': "lapack: numpy's stacked SVD instead of the library's threaded Jacobi iteration",
	'upload': "torch: host -> device copies of half a GB and more through torch instead of the library's staged copy",
	'upload_block_mb': "staging block size of that copy", 'upload_threads': "host threads filling a staging block",
	'tsv': "numpy: the command line reads / writes text with numpy.loadtxt / savetxt",
	'trace': "1: timeline of a pipelined coex call",
	'host_mirror': "1: numpy-out coex ships half of the symmetric results and mirrors them on the host",
	'exchange_chunks': "cell chunks of the pipelined exchange (default 8)", 'exchange_min_ksteps': "shortest chunk worth a launch (default 128 x 32 cells)",
	'merge_partners': "0: one Gram launch per partner block", 'force_exchange': "chunks | blocks | raw: the N > 1 exchange code on a process group of one rank",
	'rank_grace_s': "sharded CLI: seconds the survivors of a failed rank get", 'job_timeout_s': "sharded CLI: overall watchdog",
}


def _parsed():
	out = {}
	for item in os.environ.get('NRM_DEBUG', '').split(','):
		if '=' in item:
			k, v = item.split('=', 1)
			out[k.strip().lower()] = v.strip()
	return out


def debug(key, default=None):
	"""Value of a debugging switch: NRM_DEBUG="key=value,..." first, then the pre-round-5 variable NRM_KEY, then the default."""
	key = key.lower()
	assert key in DEBUG_KEYS, key
	v = _parsed().get(key)
	if v is None:
		v = os.environ.get('NRM_' + key.upper())
	return default if v is None else v
