"""Build libnormalisr_hip.so (gfx950) in-tree with hipcc.  `python -m normalisr_amd.build`."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libnormalisr_hip.so')
SOURCES = sorted(glob.glob(os.path.join(HERE, 'csrc', '*.hip')))
HEADERS = sorted(glob.glob(os.path.join(HERE, 'csrc', '*.h'))) + [os.path.join(HERE, '..', 'include', 'normalisr_hip.h')]


def hipcc_path():
	for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
		if c and os.path.exists(c):
			return c
	raise RuntimeError('hipcc not found (set HIPCC or install ROCm under /opt/rocm)')


def is_stale():
	if not os.path.exists(LIB):
		return True
	t = os.path.getmtime(LIB)
	return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
	"""Compile every HIP source for gfx950 into one shared library next to the package."""
	if not force and not is_stale():
		return LIB
	# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (with AGPR accumulators v_mfma_f64_16x16x4_f64 runs at half rate on MI355X)
	cmd = [hipcc_path(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-pthread', '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-o', LIB] + SOURCES
	if verbose:
		print(' '.join(cmd))
	r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
	if r.returncode != 0:
		raise RuntimeError('hipcc failed:\n' + r.stdout)
	return LIB


if __name__ == '__main__':
	print(build(force='--force' in sys.argv, verbose=True))
