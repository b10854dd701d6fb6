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


FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-pthread', '-mllvm', '-amdgpu-mfma-vgpr-form=1']
OBJ = os.path.join(HERE, 'csrc', '_obj')


def build(force=False, verbose=False):
	"""Compile every HIP source for gfx950 (one object per source, stale ones only, in parallel) and link them into one shared
	library next to the package."""
	if not force and not is_stale():
		return LIB
	# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (with AGPR accumulators v_mfma_f64_16x16x4_f64 runs at half rate on MI355X)
	from concurrent.futures import ThreadPoolExecutor
	hipcc = hipcc_path()
	os.makedirs(OBJ, exist_ok=True)
	newest_header = max(os.path.getmtime(f) for f in HEADERS)

	def compile_one(src):
		obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
		if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
			return obj, None
		cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
		if verbose:
			print(' '.join(cmd), flush=True)
		r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
		return obj, (r.stdout if r.returncode != 0 else None)

	with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4, 8)) as pool:
		done = list(pool.map(compile_one, SOURCES))
	for obj, err in done:
		if err is not None:
			raise RuntimeError('hipcc failed:\n' + err)
	cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', LIB] + [o for o, _ in done]
	if verbose:
		print(' '.join(cmd), flush=True)
	r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
	if r.returncode != 0:
		raise RuntimeError('hipcc (link) failed:\n' + r.stdout)
	return LIB


if __name__ == '__main__':
	print(build(force='--force' in sys.argv, verbose=True))
