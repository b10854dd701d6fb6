"""ctypes binding of libnormalisr_hip.so (include/normalisr_hip.h).  No CPU fallback: if the HIP
library is missing or a call fails, this raises."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libnormalisr_hip.so')

NRM_F32, NRM_F64 = 0, 1
NRM_TSV_I64, NRM_TSV_I32, NRM_TSV_U8 = 16, 17, 18  # integer dtypes of nrm_tsv_format
NRM_S1_COMMON, NRM_S1_SKIP = -2, -1  # cell codes of nrm_single1_stream (include/normalisr_hip.h)
DESIGN_NOTONE, DESIGN_NEG, DESIGN_GT1, DESIGN_HAS1, DESIGN_NAN = 1, 2, 4, 8, 16  # bits of nrm_design_count's d_info[2]
NRM_E_ARG, NRM_E_DEVICE, NRM_E_NUMERIC, NRM_E_UNSUPPORTED = -1, -2, -3, -4
ROW_TILE, K_TILE, PCOEF, FIX_STRIDE = 128, 16, 20, 8

_i64, _i32, _vp, _dbl = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_double


class PvaluePlan(ctypes.Structure):
	_fields_ = [('a', _dbl), ('alpha', _dbl), ('ln_front', _dbl), ('umax', _dbl), ('coef', _dbl * PCOEF)]


_SIGNATURES = {
	'nrm_version': ([], _i32),
	'nrm_last_error': ([], ctypes.c_char_p),
	'nrm_device_count': ([ctypes.POINTER(_i32)], _i32),
	'nrm_set_device': ([_i32], _i32),
	'nrm_release_cache': ([], _i32),
	'nrm_last_guard': ([ctypes.POINTER(_i64), ctypes.POINTER(_dbl)], _i32),
	'nrm_host_pin': ([_vp, _i64, _i32], _i32),
	'nrm_host_unpin': ([_vp], _i32),
	'nrm_copy_to_host': ([_vp, _vp, _i64, _vp], _i32),
	'nrm_host_alloc': ([ctypes.POINTER(_vp), _i64], _i32),
	'nrm_host_free': ([_vp], _i32),
	'nrm_fill_zero': ([_vp, _i64, _vp], _i32),
	'nrm_fill_i32': ([_vp, _i32, _i64, _vp], _i32),
	'nrm_copy_rows': ([_vp, _i64, _vp, _i64, _i64, _i64, _vp], _i32),
	'nrm_residualize': ([_vp, _i32, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i32, _vp, _i64, _i64, _vp, _vp, _vp], _i32),
	'nrm_residualize_q': ([_vp, _i32, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i32, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _i64, _vp, _vp, _vp], _i32),
	'nrm_binnet_debug_buffer': ([_vp], _i32),
	'nrm_residualize_q_chunked': ([_vp, _i32, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _i64, _vp, _vp, _vp], _i32),
	'nrm_gram_f64': ([_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _i32, _i64, _i64, _vp, _vp], _i32),
	'nrm_gram_f64_band': ([_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _i32, _i64, _i64, _i64, _i64, _vp, _vp], _i32),
	'nrm_gram_workspace_bytes': ([], _i64),
	'nrm_quant_bytes': ([_i64, _i64, _i32], _i64),
	'nrm_quantize_rows': ([_vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp], _i32),
	'nrm_gram_i8_band': ([_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i64, _i64, _vp, _vp], _i32),
	'nrm_gram_i8_chunk': ([_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i32, _i64, _i64, _i32, _i32, _vp, _vp], _i32),
	'nrm_pvalue_plan_init': ([ctypes.POINTER(PvaluePlan), _dbl], _i32),
	'nrm_pvalue_plan_init_many': ([_vp, _i64, _vp, _i64], _i32),
	'nrm_pvalue_plan_fill_many': ([_vp, _i64, _vp, _i64], _i32),
	'nrm_pvalues_from_r2': ([_vp, _i64, _dbl, _vp, _vp], _i32),
	'nrm_assoc_sweep': ([_vp, _i64, _vp, _vp, _i64, _i64, _i64, _dbl, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _dbl, _vp], _i32),
	'nrm_assoc_sweep_band': ([_vp, _i64, _vp, _vp, _i64, _i64, _i64, _dbl, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _dbl, _vp], _i32),
	'nrm_assoc_sweep_mirror': ([_vp, _i64, _vp, _vp, _i64, _i64, _i64, _dbl, _vp, _vp, _i32, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _dbl, _vp], _i32),
	'nrm_host_mirror_rows': ([_vp, _i64, _i32, _i64, _i64, _i32], _i32),
	'nrm_copy_rect_to_host': ([_vp, _i64, _vp, _i64, _i64, _i64, _vp], _i32),
	'nrm_single4_sweep': ([_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _dbl, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp], _i32),
	'nrm_gram_i8_fix_dot': ([_vp, _i64, _vp, _vp, _i64, _i64, _i32, _i64, _vp], _i32),
	'nrm_single4_sweep_guarded': ([_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _dbl, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _vp,
								  _vp, _vp, _dbl, _dbl, _i32, _dbl, _vp, _vp], _i32),
	'nrm_spd_prepare': ([_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp], _i32),
	'nrm_spd_start': ([_vp, _i64, _i32, _vp, _vp, _vp], _i32),
	'nrm_spd_transpose_residual': ([_vp, _i64, _vp, _vp, _vp, _vp], _i32),
	'nrm_spd_update': ([_vp, _vp, _i64, _vp], _i32),
	'nrm_spd_finish': ([_vp, _i64, _i64, _vp, _vp, _vp, _vp], _i32),
	'nrm_single4_design_scalars': ([_vp, _i64, _i64, _vp, _vp, _vp, _vp], _i32),
	'nrm_design_products_workspace_doubles': ([_i64, _i64], _i64),
	'nrm_design_products': ([_vp, _i32, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i32, _vp], _i32),
	'nrm_residualize_wide': ([_vp, _i32, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _i32, _vp], _i32),
	'nrm_gram_skinny': ([_vp, _i32, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _dbl, _vp, _vp], _i32),
	'nrm_gram_skinny_workspace_bytes': ([], _i64),
	'nrm_de_small_sweep': ([_vp, _vp, _vp, _i64, _i32, _vp, _i64, _i64, _i64, _dbl, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _vp], _i32),
	'nrm_single1_sweep': ([_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp], _i32),
	'nrm_de_sparse_chunk': ([], _i64),
	'nrm_de_sparse_max_covariates': ([], _i64),
	'nrm_de_sparse_fused_covariates': ([], _i64),
	'nrm_de_sparse_ct_doubles': ([_i64, _i64, _i64], _i64),
	'nrm_de_sparse': ([_vp, _i32, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _dbl, _vp, _vp], _i32),
	'nrm_design_count': ([_vp, _i32, _i64, _i64, _i64, _vp, _i64, _vp, _vp], _i32),
	'nrm_design_plan': ([_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i32),
	'nrm_design_fill': ([_vp, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp], _i32),
	'nrm_single1_select_gram_blocks': ([], _i64),
	'nrm_single1_select': ([_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i32),
	'nrm_single1_common_gram': ([_vp, _i64, _vp, _i64, _i64, _vp, _vp], _i32),
	'nrm_design_stats': ([_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp], _i32),
	'nrm_upload': ([_vp, _vp, _i64, _i32, _vp], _i32),
	'nrm_upload_release': ([], _i32),
	'nrm_host_minmax': ([_vp, _i32, _i64, _i32, _vp], _i32),
	'nrm_small_pinv': ([_vp, _i64, _i64, _dbl, _vp, _vp, _i32], _i32),
	'nrm_tsv_shape': ([_vp, _i64, _i32, _i32, _vp, _vp], _i32),
	'nrm_tsv_parse': ([_vp, _i64, _i32, _i32, _vp, _i32, _i64, _i64, _i64], _i32),
	'nrm_tsv_width': ([_i32], _i64),
	'nrm_tsv_format': ([_vp, _i32, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _i32], _i32),
	'nrm_single1_group_stats': ([_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp], _i32),
	'nrm_single1_group_info': ([_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp], _i32),
	'nrm_single1_stream': ([_vp, _i32, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _vp], _i32),
	'nrm_single1_cells': ([_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp], _i32),
	'nrm_binnet': ([_vp, _i32, _i64, _i64, _dbl, _vp, _i64, _vp, _vp, _vp], _i32),
	'nrm_binnet_rows': ([_vp, _i32, _i64, _i64, _i64, _i64, _dbl, _vp, _i64, _vp, _vp, _vp], _i32),
	'nrm_normvar_weights': ([_vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp], _i32),
	'nrm_normvar_apply': ([_vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i32, _i64, _vp, _vp], _i32),
	'nrm_normvar_device_covariates': ([], _i64),
	'nrm_normvar_exp_probe': ([_vp, _i64, _vp, _vp], _i32),
	'nrm_normvar_solve': ([_vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _dbl, _i32, _vp, _vp, _vp, _vp, _vp, _vp], _i32),
	'nrm_normvar_apply_w2': ([_vp, _i32, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _i32, _i64, _vp], _i32),
	'nrm_alpha': ([_vp, _i32, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _i32, _vp], _i32),
	'nrm_association_tests_single1_host': ([_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32], _i32),
	'nrm_association_tests_single4_host': ([_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i64, _i64, _vp, _i32, _i32, _i32, _dbl, _vp, _vp, _vp, _vp, _vp, _i32], _i32),
	'nrm_binnet_host': ([_vp, _i32, _i64, _dbl, _vp, _vp], _i32),
	'nrm_gram_host': ([_vp, _i32, _i64, _vp, _i32, _i64, _i64, _vp, _vp, _vp], _i32),
	'nrm_pvalues_host': ([_vp, _i64, _dbl, _vp], _i32),
	'nrm_normvar_host': ([_vp, _i32, _i64, _i64, _vp, _vp, _vp, _i64, _dbl, _i32, _vp, _i32, _vp], _i32),
	'nrm_small_eigvals': ([_vp, _i64, _vp], _i32),
	'nrm_association_tests_host': ([_vp, _i32, _i64, _vp, _i32, _i64, _vp, _i32, _i64, _i64, _vp, _i32, _i32, _i32,
									_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32], _i32),
}

_lib = None


def exported_symbols():
	return sorted(_SIGNATURES)


_prefer_host = False


def prefer_host_entry(on=True):
	"""The command line's default for de / coex / binnet on one GPU: numpy buffers through the library's whole-problem entries, torch not imported."""
	global _prefer_host
	prev, _prefer_host = _prefer_host, bool(on)
	return prev


def host_entry_preferred():
	v = os.environ.get('NRM_HOST_ENTRY', '')
	return v == '1' or (_prefer_host and v != '0')


def _preload_torch_hip():
	try:
		import importlib.util
		spec = importlib.util.find_spec('torch')  # (finds the package, does not import it)
		for d in (spec.submodule_search_locations or []) if spec is not None else []:
			f = os.path.join(d, 'lib', 'libamdhip64.so')
			if os.path.exists(f):
				ctypes.CDLL(f, mode=ctypes.RTLD_GLOBAL)
				return True
	except Exception:
		pass
	return False


def load():
	"""Load the HIP library; raises RuntimeError (never falls back) when it is not built."""
	global _lib
	if _lib is not None:
		return _lib
	if not os.path.exists(LIB_PATH):
		try:  # build in-tree once if a ROCm toolchain is present; never fall back to a CPU path
			from . import build as _build
			_build.build()
		except Exception as e:
			raise RuntimeError('libnormalisr_hip.so is not built ({}) and building it failed: {}. Run `python -m normalisr_amd.build` '
							   '(needs hipcc); there is no CPU fallback.'.format(LIB_PATH, e))
	# One HIP runtime per process: torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  If it is
	# loaded first our DT_NEEDED entry binds to it; the other order would load /opt/rocm's copy next to
	# torch's and the second runtime finds no device.  torch is the plumbing for device memory anyway.
	if 'torch' not in sys.modules and host_entry_preferred():
		# a process on the library's own whole-problem entries (the command line; NRM_HOST_ENTRY=1) never needs torch: its import alone is a
		# second of a 1.3 s `normalisr coex` call.  The HIP runtime torch bundles is loaded by itself instead, so that a later `import torch`
		# (a call the entries do not cover) still finds ONE runtime in the process.
		_preload_torch_hip()
	else:
		try:
			import torch  # noqa: F401
		except ImportError:
			pass
	lib = ctypes.CDLL(LIB_PATH)
	for name, (args, res) in _SIGNATURES.items():
		f = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
		f.argtypes = args
		f.restype = res
	_lib = lib
	return lib


def check(rc):
	"""Map a status code to the exception class the reference raises for the same condition."""
	if rc == 0:
		return
	msg = load().nrm_last_error().decode('utf-8', 'replace')
	if rc == NRM_E_ARG:
		raise ValueError(msg)
	if rc == NRM_E_NUMERIC:
		raise AssertionError(msg)
	if rc == NRM_E_UNSUPPORTED:
		raise NotImplementedError(msg)
	raise RuntimeError(msg)
