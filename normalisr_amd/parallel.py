"""Thread-pool map kept for API compatibility with the reference's parallel module.  The device path
does not use it: tiles are scheduled by the GPU and across GPUs by normalisr_amd.distributed."""
import logging
import multiprocessing

autocount = multiprocessing.cpu_count


def autopooler_caller(a):
	return a[0](*a[1], **a[2])


def autopooler(n, it, *a, chunksize=1, dummy=False, return_iter=False, unordered=False, **ka):
	"""Run (function, args, kwargs) triples, n at a time (0 = all cores); results in input order unless
	unordered.  Same contract as reference parallel.py:12-74."""
	if return_iter:
		raise NotImplementedError
	if n == 0:
		n = autocount()
		logging.info('Using {} threads'.format(n))
	tasks = list(it)
	assert len(tasks) > 0
	if n == 1:
		return [autopooler_caller(t) for t in tasks]
	if dummy:
		from multiprocessing.dummy import Pool
	else:
		from multiprocessing import Pool
	with Pool(n, *a, **ka) as pool:
		mapper = pool.imap_unordered if unordered else pool.imap
		return list(mapper(autopooler_caller, tasks, chunksize))


assert __name__ != "__main__"
