"""Build libnwayhip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m nway_amd.build [--force]

hipcc cross-compiles without a GPU; the built .so sits next to the sources
(git-ignored, but it travels to the GPU box with the tree).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(ROOT, 'include')
LIBRARY = os.path.join(CSRC, 'libnwayhip.so')
SOURCES = [os.path.join(CSRC, 'nwayhip.hip')]
HEADERS = [os.path.join(INCLUDE, 'nwayhip.h')] + sorted(
	os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.inc'))
ARCH = 'gfx950'


def hipcc_path():
	for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
		if cand and os.path.exists(cand):
			return cand
	raise RuntimeError('hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin)')


def is_stale():
	if not os.path.exists(LIBRARY):
		return True
	t = os.path.getmtime(LIBRARY)
	return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False):
	"""compile if the library is missing or older than its sources; returns its path"""
	if not force and not is_stale():
		return LIBRARY
	cmd = [hipcc_path(), '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-shared',
		'-I' + INCLUDE, '-I' + CSRC, '-Wall', '-Wno-unused-function'] + SOURCES + ['-o', LIBRARY + '.tmp']
	if verbose:
		print(' '.join(cmd))
	res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
	if res.returncode != 0:
		raise RuntimeError('hipcc failed:\n' + res.stdout)
	if verbose and res.stdout.strip():
		print(res.stdout)
	os.replace(LIBRARY + '.tmp', LIBRARY)
	return LIBRARY


if __name__ == '__main__':
	print(build_library(force='--force' in sys.argv, verbose=True))
