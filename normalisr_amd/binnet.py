"""Binarisation of co-expression P-value networks (mirror of the reference's binnet module, binnet.py).
`binnet` runs on the device (csrc/nrm_binnet.hip); `bh`, `nodiag`, `rediag` are small host utilities."""
import logging

import numpy as np


def nodiag(d, split=False):
	"""Off-diagonal entries of a 2-D matrix, row by row (binnet.py:4-34)."""
	d = np.asarray(d)
	assert d.ndim == 2
	k = min(d.shape)
	rows = [np.concatenate([d[i, :i], d[i, i + 1:]]) for i in range(k)] + list(d[k:])
	return rows if split else np.concatenate(rows)


def rediag(d, fill=0, shape=None):
	"""Inverse of nodiag(split=False) with `fill` on the diagonal (binnet.py:37-74)."""
	d = np.asarray(d)
	if shape is None:
		t = int(np.sqrt(d.size)) + 1
		assert t * (t - 1) == d.size
		shape = (t, t)
	assert len(shape) == 2 and shape[0] * shape[1] - min(shape) == d.size
	m = np.full(shape, fill, dtype=d.dtype)
	k = min(shape)
	off = np.ones(shape, dtype=bool)
	off[np.arange(k), np.arange(k)] = False
	m[off] = d
	return m


def bh(pv, weight=None):
	"""Benjamini-Hochberg q-values with ties and optional weights (binnet.py:77-131)."""
	pv = np.asarray(pv)
	assert pv.ndim == 1 and pv.size > 0
	assert np.isfinite(pv).all() and pv.min() >= 0 and pv.max() <= 1
	if weight is None:
		weight = np.ones(pv.size)
	else:
		weight = np.asarray(weight)
		assert weight.shape == pv.shape
		assert np.isfinite(weight).all() and weight.min() >= 0 and weight.max() > 0
	u, ids = np.unique(pv, return_inverse=True)
	if u.size == 1:
		logging.warning('Identical p-value in all entries.')
	w = np.zeros(u.size, dtype=pv.dtype)
	np.add.at(w, ids, weight.astype(pv.dtype))
	w = np.cumsum(w)
	w /= w[-1]
	with np.errstate(divide='ignore', invalid='ignore'):
		q = u / w
	q[~np.isfinite(q)] = 1
	q = np.clip(q, 0, 1)
	q = np.minimum.accumulate(q[::-1])[::-1]
	return q[ids].astype(pv.dtype, copy=False)


def binnet(net, qcut):
	"""Binarise a symmetric co-expression P-value matrix: per-row BH q-values over the off-diagonal entries,
	thresholded at qcut (binnet.py:134-173).  Returns a boolean (n_gene, n_gene) matrix, diagonal False.
	A torch CUDA tensor is accepted (and then returned) so that coex -> binnet can stay in HBM."""
	from . import _lib
	from . import engine as _engine
	on_device = hasattr(net, 'is_cuda') and net.is_cuda  # torch tensor already in HBM (e.g. coex(..., device_out=True)[0])
	if not on_device:
		net = np.asarray(net)
	assert net.ndim == 2
	nt = net.shape[0]
	if net.shape[1] != nt or nt <= 1:
		raise ValueError('Wrong shape of net or namet.')
	if qcut <= 0 or qcut >= 1:
		raise ValueError('Q-value cutoff must be between 0 and 1.')
	from .association import _use_host_entry
	if not on_device and _use_host_entry():
		# no torch in this process (or NRM_HOST_ENTRY=1): the library's whole-problem entry, numpy buffers in and out
		import ctypes
		hp = _engine.as_input(net)
		from .association import _result
		out = _result((nt, nt), np.uint8)
		total = ctypes.c_int64(-1)
		_lib.check(_lib.load().nrm_binnet_host(hp.ctypes.data_as(ctypes.c_void_p), _lib.NRM_F64 if hp.dtype == np.float64 else _lib.NRM_F32, nt, float(qcut),
											   out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(total)))
		if total.value == 0:
			raise RuntimeError('Empty binary network.')
		return out.view(np.bool_)
	eng = _engine.get_engine()
	with eng.lock:  # one call at a time per device (engine scratch, streams and guard state are shared)
		torch = eng.torch
		with torch.cuda.device(eng.device):
			d_p = net.contiguous() if on_device else eng.upload(_engine.as_input(net))
			if d_p.dtype not in (torch.float32, torch.float64):
				d_p = d_p.to(torch.float64)
			out = torch.empty((nt, nt), dtype=torch.uint8, device=eng.device)
			total = torch.zeros(1, dtype=torch.int64, device=eng.device)
			flags = torch.zeros(2, dtype=torch.int32, device=eng.device)
			_lib.check(eng.lib.nrm_binnet(d_p.data_ptr(), _lib.NRM_F64 if d_p.dtype == torch.float64 else _lib.NRM_F32, nt, d_p.stride(0),
										  float(qcut), out.data_ptr(), out.stride(0), total.data_ptr(), flags.data_ptr(), eng._stream()))
			if int(flags[0].item()):
				raise AssertionError('P-values must be finite and within [0,1] (binnet.py:151-152).')
			if int(total.item()) == 0:
				raise RuntimeError('Empty binary network.')
			return out.view(torch.bool) if on_device else eng.download(out).view(np.bool_)  # the kernel writes exact 0/1 bytes: a view, no pass over the mask


assert __name__ != "__main__"
