"""ctypes wrapper of oracle/nway_oracle.c (test / cpu-baseline infrastructure only --
never imported by nway_amd/).  Same call convention as nway_oracle.nway_match."""
import ctypes
import os
import subprocess

import numpy

import nway_oracle as _np_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libnwayoracle.so')
LIB_OMP = os.path.join(HERE, '_build', 'libnwayoracle_omp.so')  # the same source with -fopenmp
MAXCAT, MAXPAIR = 8, 28


class Table(ctypes.Structure):
	_fields_ = [('ncat', ctypes.c_int32), ('nrows', ctypes.c_int64), ('tests', ctypes.c_int64),
		('idx', ctypes.POINTER(ctypes.c_int32) * MAXCAT), ('sep', ctypes.POINTER(ctypes.c_double) * MAXPAIR),
		('sep_max', ctypes.POINTER(ctypes.c_double)), ('ncat_col', ctypes.POINTER(ctypes.c_int8)),
		('log_bf', ctypes.POINTER(ctypes.c_double)), ('log_bf_corr', ctypes.POINTER(ctypes.c_double)),
		('prior', ctypes.POINTER(ctypes.c_double)), ('dist_post', ctypes.POINTER(ctypes.c_double)),
		('p_single', ctypes.POINTER(ctypes.c_double)), ('p_any', ctypes.POINTER(ctypes.c_double)),
		('p_i', ctypes.POINTER(ctypes.c_double)), ('match_flag', ctypes.POINTER(ctypes.c_int8))]


_libs = {}


def load(build=True, omp=False):
	"""the one-thread library, or (omp=True) the OpenMP build of the same source"""
	if omp not in _libs:
		path = LIB_OMP if omp else LIB
		src = os.path.join(HERE, 'nway_oracle.c')
		if build and (not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src)):
			subprocess.check_call(['make', '-s', '-C', HERE])
		lib = ctypes.CDLL(path)
		lib.nwayo_match.restype = ctypes.c_int
		lib.nwayo_dist.restype = ctypes.c_double
		lib.nwayo_dist.argtypes = [ctypes.c_double] * 4
		lib.nwayo_threads.restype = ctypes.c_int
		lib.nwayo_set_threads.argtypes = [ctypes.c_int]
		_libs[omp] = lib
	return _libs[omp]


def visible_cores():
	"""logical CPUs this process may be scheduled on"""
	try:
		return len(os.sched_getaffinity(0))
	except AttributeError:
		return os.cpu_count() or 1


def cpu_quota():
	"""CPUs' worth of time the container's cgroup grants (cpu.max of cgroup v2, cfs quota of v1), or None without a limit"""
	try:
		quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
		if quota != 'max':
			return float(quota) / float(period)
	except (IOError, OSError, ValueError):
		pass
	try:
		quota = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
		period = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
		if quota > 0:
			return quota / period
	except (IOError, OSError, ValueError):
		pass
	return None


def host_cores():
	"""cores this process can actually USE: the CPUs it may be scheduled on, capped by the cgroup's quota.  (The GPU boxes of this
	pool show 256 logical CPUs under a quota of 16: 256 threads then share 16 CPUs' worth of time and spend it in each other's
	barriers -- the all-core leg of round 4 was measured that way.)"""
	n = visible_cores()
	q = cpu_quota()
	if q is not None:
		n = max(1, min(n, int(q + 0.999)))
	return n


def _copy(ptr, n, dtype):
	if n == 0:
		return numpy.zeros(0, dtype=dtype)
	return numpy.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def nway_match(match_tables, match_radius, prior_completeness, prob_ratio_secondary=0.5, correction='api',
		scheme=None, radius_filter=True, err_deg=None, f32_roundtrip=False, threads=1):
	"""columns as nway_oracle.nway_match; additionally '_tests' (separation evaluations).
	threads: 1 = the plain library; n > 1 (or 0 = all the cores this process may use) = the OpenMP
	build, whose result does not depend on the number of threads"""
	if threads == 1:
		lib = load()
	else:
		lib = load(omp=True)
		lib.nwayo_set_threads(int(threads) if threads > 0 else host_cores())
	k = len(match_tables)
	names = [t['name'] for t in match_tables]
	ras = [numpy.ascontiguousarray(t['ra'], dtype=float) for t in match_tables]
	decs = [numpy.ascontiguousarray(t['dec'], dtype=float) for t in match_tables]
	sigs = [numpy.ascontiguousarray(numpy.broadcast_to(numpy.asarray(t['error'], dtype=float), r.shape)) for t, r in zip(match_tables, ras)]
	err = match_radius / 60. / 60 if err_deg is None else err_deg
	if scheme is None:
		scheme = _np_oracle.choose_scheme(list(zip(ras, decs)), err)
	dens, dens_plus = _np_oracle.source_densities(match_tables)
	comp = _np_oracle.completeness_vector(prior_completeness, k)
	ptab = numpy.zeros(1 << (k - 1))
	for pattern in range(1 << (k - 1)):
		mask = numpy.array([True] + [bool((pattern >> (c - 1)) & 1) for c in range(1, k)])
		ptab[pattern] = dens[0] * numpy.prod(comp[mask]) / numpy.prod(dens_plus[mask])
	dp = ctypes.POINTER(ctypes.c_double)
	arr = lambda xs: (dp * k)(*[x.ctypes.data_as(dp) for x in xs])
	n = (ctypes.c_int64 * k)(*[len(r) for r in ras])
	out = Table()
	rc = lib.nwayo_match(ctypes.c_int(k), arr(ras), arr(decs), arr(sigs), n, ctypes.c_int(scheme), ctypes.c_int(1 if radius_filter else 0),
		ctypes.c_double(err), ctypes.c_double(match_radius), ptab.ctypes.data_as(dp), dens.ctypes.data_as(dp),
		dens_plus.ctypes.data_as(dp), ctypes.c_double(prob_ratio_secondary), ctypes.c_int((1 if correction == 'cli' else 0) | (2 if f32_roundtrip else 0)),
		ctypes.byref(out))
	if rc != 0:
		raise RuntimeError('nwayo_match failed: %d' % rc)
	M = out.nrows
	t = {}
	for c in range(k):
		t[names[c]] = _copy(out.idx[c], M, numpy.int64)
	p = 0
	for i in range(k):
		for j in range(i + 1, k):
			t['Separation_%s_%s' % (names[i], names[j])] = _copy(out.sep[p], M, float)
			p += 1
	t['Separation_max'] = _copy(out.sep_max, M, float)
	t['ncat'] = _copy(out.ncat_col, M, numpy.int64)
	t['dist_bayesfactor_uncorrected'] = _copy(out.log_bf, M, float)
	t['dist_bayesfactor'] = _copy(out.log_bf_corr, M, float)
	t['dist_post'] = _copy(out.dist_post, M, float)
	t['p_single'] = _copy(out.p_single, M, float)
	t['match_flag'] = _copy(out.match_flag, M, numpy.int64)
	t['prob_has_match'] = _copy(out.p_any, M, float)
	t['prob_this_match'] = _copy(out.p_i, M, float)
	t['_prior'] = _copy(out.prior, M, float)
	t['_tests'] = int(out.tests)
	lib.nwayo_free(ctypes.byref(out))
	return t
