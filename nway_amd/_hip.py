"""ctypes binding of libnwayhip.so (include/nwayhip.h) and device-buffer plumbing.

PyTorch-ROCm is used ONLY as the device allocator / stream provider: every buffer
crossing the C ABI is a raw device pointer (``tensor.data_ptr()``).  There is no CPU
fallback: if the HIP library or a GPU is missing, every compute entry point raises.
"""
from __future__ import annotations

import ctypes
import threading
import os
import sys

import numpy

from . import build as _build

MAXCAT = 8
MAXPAIR = 28
STATUS_WORDS = 32
ST_ROWS, ST_FLAGS, ST_REGISTRATIONS, ST_TESTS, ST_REGION_NEED, ST_SLOT_NEED = 0, 1, 2, 3, 4, 5
LINK_SLOTS_CAP = 128
LINK_SLOTS_MAX_FUSED = 8  # most slots of the one-launch fused tails (sparse.inc: LINK_SLOTS_MAX)
ST_SURVIVORS, ST_PAIRS, ST_NOTFLAT = 8, 16, 24
FLAG_PAIR_OVERFLOW, FLAG_ROW_OVERFLOW, FLAG_REG_OVERFLOW, FLAG_SLOT_OVERFLOW, FLAG_LOOKBACK, FLAG_QUAD_DEEP = 1, 2, 4, 8, 16, 64
PATH_GENERAL, PATH_SPARSE, PATH_HYBRID = 0, 1, 2
SCHEME_FLAT, SCHEME_SPHERE = 0, 1
STAGES = 8
STAGE_NAMES = ['register', 'sweep', 'pairs', 'lists', 'expand', 'rows', 'groups']
CORRECTION_NONE, CORRECTION_CLI = 0, 1
DISABLE_DENSE3, DISABLE_HYBRID, DISABLE_FUSED_CORRECTION, DISABLE_ONE_SWEEP, DISABLE_QUAD3 = 1, 2, 4, 8, 16
ENABLE_QUAD3 = 2
ABI_VERSION = 3  # include/nwayhip.h: NWAYHIP_ABI_VERSION
MAXZONES = 64
DESC_WORDS = 8
DESC_NAMES = ['path', 'link_slots', 'direct_log2', 'sweep', 'tail', 'fold_log2', 'one_sweep', 'reserved']
SWEEP_GENERAL, SWEEP_LDS, SWEEP_BIG = 0, 1, 2
SWEEP_NAMES = ['general', 'lds', 'big']
TAIL_GENERAL, TAIL_SPARSE2, TAIL_DENSE2, TAIL_SPARSEK, TAIL_DENSE3, TAIL_HYBRID = 0, 1, 2, 3, 4, 5
TAIL_NAMES = ['general', 'sparse2', 'dense2', 'sparsek', 'dense3', 'hybrid', 'quad3']


class NwayHipError(RuntimeError):
	pass


class Catalogue(ctypes.Structure):
	_fields_ = [('ra', ctypes.c_void_p), ('dec', ctypes.c_void_p), ('sigma', ctypes.c_void_p),
		('sigma_const', ctypes.c_double), ('n', ctypes.c_int64)]


class MatchParams(ctypes.Structure):
	_fields_ = [('ncat', ctypes.c_int32), ('scheme', ctypes.c_int32), ('radius_filter', ctypes.c_int32),
		('correction', ctypes.c_int32), ('finalize', ctypes.c_int32), ('link_slots', ctypes.c_int32),
		('err_deg', ctypes.c_double), ('radius_arcsec', ctypes.c_double), ('prob_ratio_secondary', ctypes.c_double),
		('dens', ctypes.c_double * MAXCAT), ('dens_plus', ctypes.c_double * MAXCAT),
		('prior_table', ctypes.c_double * (1 << (MAXCAT - 1))),
		('sphere_cell_factor', ctypes.c_double), ('bitmap_bits', ctypes.c_int64), ('table_slots', ctypes.c_int64),
		('link_region_min', ctypes.c_int64), ('f32_roundtrip', ctypes.c_int64),
		('direct_log2', ctypes.c_int32), ('fold_log2', ctypes.c_int32), ('disable', ctypes.c_int32), ('enable', ctypes.c_int32)]


class Table(ctypes.Structure):
	_fields_ = [('capacity', ctypes.c_int64), ('idx', ctypes.c_void_p * MAXCAT), ('sep', ctypes.c_void_p * MAXPAIR),
		('sep_max', ctypes.c_void_p), ('ncat', ctypes.c_void_p), ('log_bf', ctypes.c_void_p),
		('log_bf_corrected', ctypes.c_void_p), ('prior', ctypes.c_void_p), ('dist_post', ctypes.c_void_p),
		('p_single', ctypes.c_void_p), ('p_any', ctypes.c_void_p), ('p_i', ctypes.c_void_p),
		('match_flag', ctypes.c_void_p), ('group_start', ctypes.c_void_p)]


class Split(ctypes.Structure):
	"""nwayhip_split: secondary-split mode (several GPUs on one job)"""
	_fields_ = [('world', ctypes.c_int32), ('rank', ctypes.c_int32), ('d_bounds', ctypes.c_void_p),
		('h_p_lo', ctypes.c_int64), ('h_p_hi', ctypes.c_int64), ('slice_offset', ctypes.c_int64 * MAXCAT),
		('capacity', ctypes.c_int64), ('d_export', ctypes.c_void_p), ('d_import', ctypes.c_void_p)]


class ZoneRun(ctypes.Structure):
	"""nwayhip_zone_run: what nwayhip_match_enqueue takes, per zone of a launch set"""
	_fields_ = [('h_cats', ctypes.POINTER(Catalogue)), ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t),
		('h_table', ctypes.POINTER(Table)), ('d_status', ctypes.c_void_p)]


# every symbol include/nwayhip.h declares: (restype, argtypes)
_vp, _i64, _i32, _dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double
SYMBOLS = {
	'nwayhip_version': (ctypes.c_int, []),
	'nwayhip_last_error': (ctypes.c_char_p, []),
	'nwayhip_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
	'nwayhip_dist': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
	'nwayhip_dist_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
	'nwayhip_log_bf': (ctypes.c_int, [_i32, _i64, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp]),
	'nwayhip_log_bf_elliptical': (ctypes.c_int, [_i32, _i64, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
		ctypes.POINTER(_vp), _vp, _i32, _vp]),
	'nwayhip_offsets': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
	'nwayhip_posterior': (ctypes.c_int, [_i32, _vp, _vp, _i64, _vp, _vp]),
	'nwayhip_fastmath_probe': (ctypes.c_int, [_i32, _vp, _vp, _i64, _vp, _vp, _vp]),
	'nwayhip_plan_create': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(MatchParams), ctypes.POINTER(_i64), _i64, _i64]),
	'nwayhip_plan_destroy': (ctypes.c_int, [_vp]),
	'nwayhip_plan_workspace_bytes': (ctypes.c_size_t, [_vp]),
	'nwayhip_plan_table_slots': (ctypes.c_int64, [_vp]),
	'nwayhip_plan_link_slots': (ctypes.c_int32, [_vp]),
	'nwayhip_plan_path': (ctypes.c_int32, [_vp]),
	'nwayhip_plan_describe': (ctypes.c_int, [_vp, ctypes.POINTER(_i32)]),
	'nwayhip_plan_split_capable': (ctypes.c_int32, [_vp]),
	'nwayhip_match_enqueue': (ctypes.c_int, [_vp, ctypes.POINTER(Catalogue), _vp, ctypes.c_size_t, ctypes.POINTER(Table), _vp, _vp]),
	'nwayhip_zones_create': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i32]),
	'nwayhip_zones_destroy': (ctypes.c_int, [_vp]),
	'nwayhip_zones_args_bytes': (ctypes.c_size_t, [_vp]),
	'nwayhip_zones_enqueue': (ctypes.c_int, [_vp, ctypes.POINTER(ZoneRun), _vp, ctypes.c_size_t, _vp]),
	'nwayhip_zones_batched': (ctypes.c_int32, [_vp]),
	'nwayhip_zones_set_registration': (ctypes.c_int, [_vp, _i32]),
	'nwayhip_split_buffer_bytes': (ctypes.c_size_t, [_vp, _i32, _i64]),
	'nwayhip_split_front_enqueue': (ctypes.c_int, [_vp, ctypes.POINTER(Catalogue), _vp, ctypes.c_size_t, ctypes.POINTER(Split), _vp, _vp]),
	'nwayhip_split_back_enqueue': (ctypes.c_int, [_vp, ctypes.POINTER(Catalogue), _vp, ctypes.c_size_t, ctypes.POINTER(Split), ctypes.POINTER(Table), _vp, _vp]),
	'nwayhip_plan_profile': (ctypes.c_int, [_vp, ctypes.c_uint32]),
	'nwayhip_plan_profile_read': (ctypes.c_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_dbl)]),
	'nwayhip_plan_profile_stride': (ctypes.c_int, [_vp, _i32]),
	'nwayhip_plan_profile_samples': (ctypes.c_int, [_vp, _i32, ctypes.POINTER(_dbl), _i64, ctypes.POINTER(_i64)]),
	'nwayhip_group_stats': (ctypes.c_int, [_i64, _i64, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, _vp, _vp]),
	'nwayhip_bias_lookup': (ctypes.c_int, [_i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
	'nwayhip_catalogue_extent': (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp]),
	'nwayhip_read_probe': (ctypes.c_int, [_vp, _vp, _i64, _vp, _i32, _vp]),
	'nwayhip_comm_unique_id': (ctypes.c_int, [_vp]),
	'nwayhip_comm_init': (ctypes.c_int, [ctypes.POINTER(_vp), _i32, _i32, _vp]),
	'nwayhip_comm_destroy': (ctypes.c_int, [_vp]),
	'nwayhip_comm_world': (ctypes.c_int32, [_vp]),
	'nwayhip_comm_rank': (ctypes.c_int32, [_vp]),
	'nwayhip_comm_allgatherv_f64': (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_i64), _vp, _vp]),
	'nwayhip_comm_exchange': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, _vp]),
}

_lib = None


def library_path():
	"""the in-tree build, or the prebuilt library named by $NWAYHIP_LIBRARY"""
	return os.environ.get('NWAYHIP_LIBRARY') or _build.LIBRARY


def load():
	"""dlopen libnwayhip.so and bind the prototypes (works without a GPU)."""
	global _lib
	if _lib is not None:
		return _lib
	# PyTorch-ROCm ships its own copy of the HIP runtime: it has to be in the process BEFORE
	# libnwayhip.so is opened, so that the library binds to that same runtime (opened the other
	# way round, the process ends up with two runtimes and this one sees no device)
	torch()
	path = library_path()
	if not os.path.exists(path):
		raise NwayHipError('HIP library %s is not built; run "python -m nway_amd.build" (needs hipcc). '
			'There is no CPU fallback.' % path)
	lib = ctypes.CDLL(path)
	# (the version first: an older library lacks symbols the binding below would ask for)
	lib.nwayhip_version.restype = ctypes.c_int
	lib.nwayhip_version.argtypes = []
	if lib.nwayhip_version() != ABI_VERSION:
		raise NwayHipError('ABI version mismatch: library %s has %d, binding %d (rebuild: python -m nway_amd.build --force)' % (path, lib.nwayhip_version(), ABI_VERSION))
	for name, (restype, argtypes) in SYMBOLS.items():
		fn = getattr(lib, name)
		fn.restype = restype
		fn.argtypes = argtypes
	_lib = lib
	return lib


def check(rc):
	if rc != 0:
		raise NwayHipError(load().nwayhip_last_error().decode('utf-8', 'replace'))


def device_count():
	n = ctypes.c_int(0)
	rc = load().nwayhip_device_count(ctypes.byref(n))
	return n.value if rc == 0 else 0


_torch = None


def torch():
	global _torch
	if _torch is None:
		import torch as t
		_torch = t
	return _torch


def require_device(device=None):
	"""torch.device of the GPU to use; raises when there is none (no CPU fallback)."""
	t = torch()
	if not t.cuda.is_available() or device_count() < 1:
		raise NwayHipError('no AMD GPU visible: the nway hot path runs only as HIP kernels (no CPU fallback)')
	if device is None:
		return t.device('cuda', t.cuda.current_device())
	return t.device(device)


UPLOAD_PIN_BYTES = 1 << 20     # columns of at least this size are uploaded from page-locked memory
UPLOAD_CHUNK_BYTES = 16 << 20  # staging buffers of the fallback path
upload_mode = {'last': None}   # how the last large column travelled: 'registered' | 'staged' (bench.py reports it)


def _upload_large(a, out):
	"""host array -> device tensor ``out`` (same shape and dtype) at DMA speed.  A pageable source goes
	through the runtime's own bounce buffers at ~1 GB/s on this stack; page-locking the caller's array
	in place (hipHostRegister, undone afterwards) lets the copy engine read it directly.  Where the
	runtime refuses, two pinned staging buffers alternate: the host copies chunk i + 1 into one while
	the engine drains chunk i from the other."""
	t = torch()
	src = t.from_numpy(a)
	rt = t.cuda.cudart()
	registered = False
	try:
		if os.environ.get('NWAY_UPLOAD', '') != 'staged':  # (development: the staged path on a stack that registers)
			rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
			registered = int(rc) == 0
	except Exception:
		registered = False
	if registered:
		try:
			out.copy_(src, non_blocking=True)
			t.cuda.current_stream(out.device).synchronize()
		finally:
			rt.cudaHostUnregister(a.ctypes.data)
		upload_mode['last'] = 'registered'
		return out
	flat_src, flat_out = src.reshape(-1), out.reshape(-1)
	per = max(1, UPLOAD_CHUNK_BYTES // a.itemsize)
	stage = [t.empty(per, dtype=src.dtype, pin_memory=True) for _ in range(2)]
	done = [None, None]
	stream = t.cuda.current_stream(out.device)
	for i, lo in enumerate(range(0, flat_src.shape[0], per)):
		hi = min(lo + per, flat_src.shape[0])
		b = i & 1
		if done[b] is not None:
			done[b].synchronize()
		stage[b][:hi - lo].copy_(flat_src[lo:hi])
		flat_out[lo:hi].copy_(stage[b][:hi - lo], non_blocking=True)
		done[b] = t.cuda.Event()
		done[b].record(stream)
	stream.synchronize()
	upload_mode['last'] = 'staged'
	return out


_download_stage = {}
_download_lock = threading.Lock()  # (the staging buffers are shared: two threads downloading at once would overwrite each other's)


def to_host(tensor):
	"""device tensor -> numpy array of its own.  Large columns come down through a page-locked staging buffer kept for the
	purpose (the runtime otherwise page-locks the fresh destination on the fly for every copy); NWAY_DOWNLOAD=direct: the
	runtime's own path"""
	t = torch()
	nbytes = tensor.numel() * tensor.element_size()
	if not tensor.is_cuda or nbytes < (64 << 10) or os.environ.get('NWAY_DOWNLOAD', '') == 'direct':
		return tensor.cpu().numpy()
	src = tensor.contiguous()
	with _download_lock:
		stage = _download_stage.get(src.dtype)
		if stage is None or stage.numel() < src.numel():
			stage = t.empty(max(src.numel(), (4 << 20) // src.element_size()), dtype=src.dtype, pin_memory=True)
			_download_stage[src.dtype] = stage
		view = stage[:src.numel()]
		view.copy_(src.reshape(-1), non_blocking=True)
		t.cuda.current_stream(src.device).synchronize()
		return view.numpy().reshape(tuple(src.shape)).copy()


def to_device(array, device, dtype=None):
	"""numpy array or torch tensor -> contiguous device tensor (float64 by default)"""
	t = torch()
	dtype = dtype or t.float64
	if isinstance(array, t.Tensor):
		return array.to(device=device, dtype=dtype).contiguous()
	a = numpy.ascontiguousarray(numpy.asarray(array), dtype={t.float64: numpy.float64, t.float32: numpy.float32, t.int32: numpy.int32, t.int64: numpy.int64}[dtype])
	if a.nbytes >= UPLOAD_PIN_BYTES and t.device(device).type == 'cuda':
		if not a.flags.writeable:
			a = a.copy()  # (torch.from_numpy wants a writable buffer)
		return _upload_large(a, t.empty(a.shape, dtype=dtype, device=device))
	return t.from_numpy(a).to(device)


UPLOAD_STAGE_COLUMN_BYTES = UPLOAD_PIN_BYTES   # columns below this size travel TOGETHER through one page-locked staging buffer (8 MB tried: the host copy of MB-size columns costs more than page-locking them in place -- C1' 1.0 -> 1.5 ms) ...
UPLOAD_STAGE_TOTAL_BYTES = 64 << 20   # ... of at most this size (beyond it a host copy costs more than page-locking in place)
_upload_stage = {}
_upload_lock = threading.Lock()


def _upload_staged(columns, dev):
	"""float64 host columns -> views of ONE device buffer, through ONE page-locked staging buffer and ONE transfer (the copy is
	enqueued, not waited for: the caller synchronises once).  A column of a few hundred KB costs 0.1-0.6 ms on the runtime's pageable
	path and as much to page-lock in place; thirteen of them were 4 of the 8 ms of a 3-way match with magnitude priors (round 5)."""
	t = torch()
	offsets, total = [], 0
	for a in columns:
		offsets.append(total)
		total += (a.nbytes + 255) // 256 * 256
	with _upload_lock:
		stage = _upload_stage.get('host')
		if stage is None or stage.numel() < total:
			stage = t.empty(max(total, 4 << 20), dtype=t.uint8, pin_memory=True)
			_upload_stage['host'] = stage
		host = stage.numpy()
		for a, off in zip(columns, offsets):
			host[off:off + a.nbytes] = a.reshape(-1).view(numpy.uint8)
		buf = t.empty(total, dtype=t.uint8, device=dev)
		buf.copy_(stage[:total], non_blocking=True)
		# (the staging buffer is free again once the copy has run: the caller's one synchronisation, inside the lock's owner)
		t.cuda.current_stream(dev).synchronize()
	return [buf[off:off + a.nbytes].view(t.float64).reshape(a.shape) for a, off in zip(columns, offsets)]


def upload_columns(arrays, device):
	"""several host columns -> float64 device tensors with ONE synchronisation: the large ones are page-locked in place, the small
	ones go together through one page-locked staging buffer (``_upload_staged``), all copies are issued, the stream is waited for
	once, the arrays are released.  Tensors already on a device pass through ``to_device``."""
	t = torch()
	dev = t.device(device)
	rt = t.cuda.cudart()
	outs, registered = [], []
	small = []  # (position in outs, host column)
	try:
		for array in arrays:
			if isinstance(array, t.Tensor) or dev.type != 'cuda':
				outs.append(to_device(array, device))
				continue
			a = numpy.ascontiguousarray(numpy.asarray(array), dtype=numpy.float64)
			if os.environ.get('NWAY_UPLOAD', '') == 'staged':
				outs.append(to_device(a, device))
				continue
			if a.nbytes < UPLOAD_STAGE_COLUMN_BYTES and sum(x.nbytes for _, x in small) + a.nbytes <= UPLOAD_STAGE_TOTAL_BYTES:
				small.append((len(outs), a))
				outs.append(None)
				continue
			if a.nbytes < UPLOAD_PIN_BYTES:
				outs.append(to_device(a, device))
				continue
			if not a.flags.writeable:
				a = a.copy()
			ok = False
			try:
				ok = int(rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)) == 0
			except Exception:
				ok = False
			if not ok:
				outs.append(to_device(a, device))
				continue
			registered.append(a)
			out = t.empty(a.shape, dtype=t.float64, device=dev)
			out.copy_(t.from_numpy(a), non_blocking=True)
			outs.append(out)
		if small:
			for (at, _), dev_col in zip(small, _upload_staged([x for _, x in small], dev)):
				outs[at] = dev_col
			upload_mode['last'] = 'staged-small'
		if registered:
			t.cuda.current_stream(dev).synchronize()
			upload_mode['last'] = 'registered'
	finally:
		for a in registered:
			rt.cudaHostUnregister(a.ctypes.data)
	return outs


def current_stream_ptr(device):
	return ctypes.c_void_p(torch().cuda.current_stream(device).cuda_stream)


def ptr(tensor):
	return ctypes.c_void_p(tensor.data_ptr()) if tensor is not None else ctypes.c_void_p(0)


def pair_columns(ncat):
	return [(i, j) for i in range(ncat) for j in range(i + 1, ncat)]


def catalogue_extent(ra, dec):
	"""(min ra, max ra, max |dec|, #NaN) of device-resident columns (k_extent); synchronises"""
	t = torch()
	out = t.empty(4, dtype=t.float64, device=ra.device)
	check(load().nwayhip_catalogue_extent(ptr(ra), ptr(dec), int(ra.shape[0]), ptr(out), current_stream_ptr(ra.device)))
	lo, hi, absdec, nnan = out.cpu().numpy()
	return float(lo), float(hi), float(absdec), int(nnan)


def catalogue_extents(catalogues):
	"""``catalogue_extent`` of several device catalogues with one synchronisation"""
	t = torch()
	if not catalogues:
		return []
	dev = catalogues[0].ra.device
	out = t.empty((len(catalogues), 4), dtype=t.float64, device=dev)
	for i, c in enumerate(catalogues):
		check(load().nwayhip_catalogue_extent(ptr(c.ra), ptr(c.dec), int(c.n), ptr(out[i]), current_stream_ptr(dev)))
	return [(float(lo), float(hi), float(absdec), int(nnan)) for lo, hi, absdec, nnan in out.cpu().numpy()]


def scheme_from_extents(extents, err):
	"""flat-cell condition of fastskymatch.py:94-98 evaluated on per-catalogue extents"""
	for lo, hi, absdec, nnan in extents:
		if not (err < 1 and nnan == 0 and lo > 10 * err and hi < 360 - 10 * err and absdec < 45):  # (a NaN coordinate fails the host test too)
			return SCHEME_SPHERE
	return SCHEME_FLAT


COMM_ID_BYTES = 128


class RcclComm(object):
	"""RCCL communicator behind the C ABI (nwayhip_comm_*): one per process / GPU.  ``id_bytes``: the 128 bytes rank 0 got
	from ``RcclComm.unique_id()``, handed over by the caller's side channel (nway_amd.distributed broadcasts them)."""

	@staticmethod
	def unique_id():
		buf = (ctypes.c_char * COMM_ID_BYTES)()
		check(load().nwayhip_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
		return bytes(buf.raw)

	def __init__(self, world, rank, id_bytes, device):
		self.lib = load()
		self.device = require_device(device)
		self.world, self.rank = int(world), int(rank)
		if len(id_bytes) != COMM_ID_BYTES:
			raise ValueError('the communicator id has %d bytes' % COMM_ID_BYTES)
		handle = ctypes.c_void_p(0)
		buf = ctypes.create_string_buffer(bytes(id_bytes), COMM_ID_BYTES)
		with torch().cuda.device(self.device):
			check(self.lib.nwayhip_comm_init(ctypes.byref(handle), self.world, self.rank, ctypes.cast(buf, ctypes.c_void_p)))
		self.handle = handle

	def allgatherv(self, tensor, counts):
		"""every rank's 1-D float64 slice -> the whole column (rank order) on every GPU; counts: rows of every rank"""
		t = torch()
		if tensor.dtype != t.float64:
			raise TypeError('RcclComm.allgatherv moves float64 columns, got %s' % tensor.dtype)
		full = t.empty(int(sum(counts)), dtype=t.float64, device=self.device)
		if full.numel() == 0:
			return full  # (nothing to move: a NULL receive buffer is not an argument the library accepts)
		arr = (ctypes.c_int64 * self.world)(*[int(c) for c in counts])
		src = tensor.contiguous()
		check(self.lib.nwayhip_comm_allgatherv_f64(self.handle, ptr(src) if int(counts[self.rank]) > 0 else None, arr, ptr(full), current_stream_ptr(self.device)))
		return full

	def exchange(self, export, imported):
		"""block r of ``export`` to rank r, block s of ``imported`` from rank s (uint8 tensors of world equal blocks)"""
		nbytes = int(export.numel()) * export.element_size()
		if nbytes % self.world or nbytes != int(imported.numel()) * imported.element_size():
			raise ValueError('exchange buffers must hold one equal block per rank')
		check(self.lib.nwayhip_comm_exchange(self.handle, ptr(export), ptr(imported), nbytes // self.world, current_stream_ptr(self.device)))

	def close(self):
		if getattr(self, 'handle', None):
			self.lib.nwayhip_comm_destroy(self.handle)
			self.handle = None

	def __del__(self):
		try:
			self.close()
		except Exception:
			pass


class DeviceCatalogue(object):
	"""ra/dec (deg) and positional error (arcsec) columns resident in HBM."""

	def __init__(self, ra, dec, error, device):
		self.ra = to_device(ra, device)
		self.dec = to_device(dec, device)
		if numpy.ndim(error) == 0 and not isinstance(error, torch().Tensor):
			self.sigma = None
			self.sigma_const = float(error)
		else:
			self.sigma = to_device(error, device)
			self.sigma_const = 0.0
		self.n = int(self.ra.shape[0])
		if self.dec.shape[0] != self.n or (self.sigma is not None and self.sigma.shape[0] != self.n):
			raise ValueError('catalogue columns differ in length')

	def struct(self):
		return Catalogue(ptr(self.ra), ptr(self.dec), ptr(self.sigma), self.sigma_const, self.n)

	@classmethod
	def from_columns(cls, columns, device):
		"""[(ra, dec, error), ...] -> catalogues, all columns uploaded behind one synchronisation (upload_columns)"""
		flat, scalar = [], []
		for ra, dec, error in columns:
			is_scalar = numpy.ndim(error) == 0 and not isinstance(error, torch().Tensor)
			scalar.append(is_scalar)
			flat += [ra, dec] + ([] if is_scalar else [error])
		dev = upload_columns(flat, device)
		out, at = [], 0
		for (ra, dec, error), is_scalar in zip(columns, scalar):
			c = cls.__new__(cls)
			c.ra, c.dec = dev[at], dev[at + 1]
			at += 2
			if is_scalar:
				c.sigma, c.sigma_const = None, float(error)
			else:
				c.sigma, c.sigma_const = dev[at], 0.0
				at += 1
			c.n = int(c.ra.shape[0])
			if c.dec.shape[0] != c.n or (c.sigma is not None and c.sigma.shape[0] != c.n):
				raise ValueError('catalogue columns differ in length')
			out.append(c)
		return out


class MatchPlan(object):
	"""Parameters + capacities + device buffers for repeated runs of the match pipeline."""

	def table_slots(self):
		return self._table_slots

	def __init__(self, sizes, params, cap_pairs, cap_rows, device, lean=False):
		"""lean: allocate only the columns somebody reads afterwards -- ``log_bf_corrected`` is the
		``log_bf`` tensor itself unless the script's correction runs, and ``prior`` is left out on
		the sparse path with finalize (the fused tail consumes it in registers)"""
		t = torch()
		self.lib = load()
		self.device = require_device(device)
		self.ncat = len(sizes)
		self.sizes = [int(n) for n in sizes]
		self.params = params
		self.cap_pairs = int(cap_pairs)
		self.cap_rows = int(cap_rows)
		handle = ctypes.c_void_p(0)
		n_arr = (ctypes.c_int64 * self.ncat)(*self.sizes)
		check(self.lib.nwayhip_plan_create(ctypes.byref(handle), ctypes.byref(params), n_arr, self.cap_pairs, self.cap_rows))
		self.handle = handle
		self.workspace_bytes = int(self.lib.nwayhip_plan_workspace_bytes(handle))
		self._table_slots = int(self.lib.nwayhip_plan_table_slots(handle))
		self.link_slots = int(self.lib.nwayhip_plan_link_slots(handle))
		self.sparse = self.link_slots > 0   # the sparse front (its overflows send a run to the general path)
		self.path = int(self.lib.nwayhip_plan_path(handle))
		self.fused = self.path == PATH_SPARSE  # a fused tail reports exact row counts (run_plan)
		desc = (ctypes.c_int32 * DESC_WORDS)()
		check(self.lib.nwayhip_plan_describe(handle, desc))
		self.description = dict(zip(DESC_NAMES, [int(x) for x in desc]))
		self.description['sweep'] = SWEEP_NAMES[self.description['sweep']]
		self.description['tail'] = TAIL_NAMES[self.description['tail']]
		self.split_capable = bool(self.lib.nwayhip_plan_split_capable(handle))
		self.attempts = 1  # (run_plan: enqueues it took to settle the capacities)
		self.lean = bool(lean)
		skip = set()
		if lean and params.correction == CORRECTION_NONE:
			skip.add('log_bf_corrected')
		if lean and self.fused and params.finalize:
			skip.add('prior')
		with t.cuda.device(self.device):
			self.workspace = t.empty(self.workspace_bytes + 256, dtype=t.uint8, device=self.device)
			self.status = t.zeros(STATUS_WORDS, dtype=t.int64, device=self.device)
			cap = self.cap_rows
			# the columns of one type are the rows of ONE tensor (stride: the capacity rounded up to 64 rows, every column
			# 256-byte aligned): what a caller downloads is then three strided copies and one transfer (download_table)
			stride = (cap + 63) // 64 * 64
			f64_names = ['sep%d' % i for i in range(len(pair_columns(self.ncat)))] + [n for n in
				('sep_max', 'log_bf', 'log_bf_corrected', 'prior', 'dist_post', 'p_single', 'p_any', 'p_i') if n not in skip]
			self.block_f64 = t.empty((len(f64_names), stride), dtype=t.float64, device=self.device)
			self.block_i32 = t.empty((self.ncat, stride), dtype=t.int32, device=self.device)
			self.block_i8 = t.empty((2, stride), dtype=t.int8, device=self.device)
			self.f64_row = dict((n, i) for i, n in enumerate(f64_names))
			self.cols = {}
			self.cols['idx'] = [self.block_i32[c, :cap] for c in range(self.ncat)]
			self.cols['sep'] = [self.block_f64[self.f64_row['sep%d' % i], :cap] for i in range(len(pair_columns(self.ncat)))]
			for name in ('sep_max', 'log_bf', 'log_bf_corrected', 'prior', 'dist_post', 'p_single', 'p_any', 'p_i'):
				self.cols[name] = None if name in skip else self.block_f64[self.f64_row[name], :cap]
			self.cols['ncat'] = self.block_i8[0, :cap]
			self.cols['match_flag'] = self.block_i8[1, :cap]
			self.cols['group_start'] = t.empty(self.sizes[0] + 1, dtype=t.int64, device=self.device)
			self.device_bytes = (self.workspace.numel() + self.block_f64.numel() * 8 + self.block_i32.numel() * 4 + self.block_i8.numel()
				+ self.cols['group_start'].numel() * 8)
		tab = Table()
		tab.capacity = cap
		for c in range(self.ncat):
			tab.idx[c] = self.cols['idx'][c].data_ptr()
		for p in range(len(self.cols['sep'])):
			tab.sep[p] = self.cols['sep'][p].data_ptr()
		for name in ('sep_max', 'ncat', 'log_bf', 'log_bf_corrected', 'prior', 'dist_post', 'p_single', 'p_any', 'p_i', 'match_flag', 'group_start'):
			setattr(tab, name, self.cols[name].data_ptr() if self.cols[name] is not None else None)
		if 'log_bf_corrected' in skip:
			self.cols['log_bf_corrected'] = self.cols['log_bf']  # the same values: one tensor
			self.f64_row['log_bf_corrected'] = self.f64_row['log_bf']
		self.table_struct = tab
		ws = self.workspace.data_ptr()
		self.ws_ptr = (ws + 255) // 256 * 256
		self.ws_len = self.workspace_bytes + 256 - (self.ws_ptr - ws)

	def enqueue(self, catalogues, stream=None):
		"""enqueue one pass of the whole pipeline; does not synchronise"""
		cats = (Catalogue * self.ncat)(*[c.struct() for c in catalogues])
		s = stream if stream is not None else current_stream_ptr(self.device)
		check(self.lib.nwayhip_match_enqueue(self.handle, cats, ctypes.c_void_p(self.ws_ptr), self.ws_len,
			ctypes.byref(self.table_struct), ptr(self.status), s))

	def split_buffer_bytes(self, world, capacity):
		return int(self.lib.nwayhip_split_buffer_bytes(self.handle, world, capacity))

	def split_front(self, catalogues, split, stream=None):
		"""secondary-split mode, first half: register all primaries, sweep the own slices, export the
		candidates (``split``: a ``Split``); the caller's all-to-all comes next"""
		cats = (Catalogue * self.ncat)(*[c.struct() for c in catalogues])
		s = stream if stream is not None else current_stream_ptr(self.device)
		check(self.lib.nwayhip_split_front_enqueue(self.handle, cats, ctypes.c_void_p(self.ws_ptr), self.ws_len, ctypes.byref(split), ptr(self.status), s))

	def split_back(self, catalogues, split, stream=None):
		"""second half: what the peers exported to this rank -> links of its own primaries -> table"""
		cats = (Catalogue * self.ncat)(*[c.struct() for c in catalogues])
		s = stream if stream is not None else current_stream_ptr(self.device)
		check(self.lib.nwayhip_split_back_enqueue(self.handle, cats, ctypes.c_void_p(self.ws_ptr), self.ws_len, ctypes.byref(split),
			ctypes.byref(self.table_struct), ptr(self.status), s))

	def profile(self, stage_mask, every=1):
		"""bracket the stages in ``stage_mask`` with HIP events on the pipeline's stream; ``every``:
		only every n-th launch of a single-launch stage carries its event pair"""
		check(self.lib.nwayhip_plan_profile_stride(self.handle, every))
		check(self.lib.nwayhip_plan_profile(self.handle, stage_mask))

	def profile_samples(self, stage):
		"""the individual durations (ms) of the bracketed launches of one stage recorded so far (before profile_read resets them)"""
		cap = 512
		ms = (ctypes.c_double * cap)()
		n = ctypes.c_int64(0)
		check(self.lib.nwayhip_plan_profile_samples(self.handle, stage, ms, cap, ctypes.byref(n)))
		return list(ms[:n.value])

	def profile_read(self):
		"""(launch groups, summed ms) per stage since the last call; waits for the events"""
		n = (ctypes.c_int64 * STAGES)()
		ms = (ctypes.c_double * STAGES)()
		check(self.lib.nwayhip_plan_profile_read(self.handle, n, ms))
		return list(n), list(ms)

	def read_status(self):
		"""synchronises; returns the status words as numpy int64"""
		return self.status.cpu().numpy()

	def download_table(self, nrows, f64_columns, with_idx=True, with_small=True):
		"""The first ``nrows`` rows of the table on the host with ONE transfer and ONE synchronisation: the index columns
		(as int64, what the reference's table holds), ncat and match_flag (int64) and the float columns named in
		``f64_columns`` (cols keys; 'sep<i>' for the separation columns; a name may repeat: it then gets memory of its own)
		are packed on the device -- one strided copy per type -- and come down into one page-locked buffer, of which the
		returned arrays are views (no copy on the host; the buffer lives as long as one of them does and goes back to
		torch's pinned-memory cache afterwards).  Returns (idx list, (ncat, match_flag) or None, list of float columns)."""
		self.check_live()
		t = torch()
		n = int(nrows)
		ni = self.ncat if with_idx else 0
		ns = 2 if with_small else 0
		nf = len(f64_columns)
		words = (ni + ns + nf) * n
		if words == 0:
			return [numpy.zeros(0, dtype=numpy.int64)] * ni, ((numpy.zeros(0, dtype=numpy.int64),) * 2 if ns else None), [numpy.zeros(0)] * nf
		with t.cuda.device(self.device):
			pack = t.empty(words, dtype=t.int64, device=self.device)
			if ni:
				pack[:ni * n].view(ni, n).copy_(self.block_i32[:, :n])   # (int32 -> int64 in the copy)
			if ns:
				pack[ni * n:(ni + ns) * n].view(ns, n).copy_(self.block_i8[:, :n])
			if nf:
				rows = t.tensor([self.f64_row[c] for c in f64_columns], dtype=t.int64, device=self.device)
				t.index_select(self.block_f64[:, :n], 0, rows, out=pack[(ni + ns) * n:].view(t.float64).view(nf, n))
			if os.environ.get('NWAY_DOWNLOAD', '') == 'direct':
				flat = pack.cpu().numpy()  # (development: the runtime's own pageable path, tools/dev/fault_study.sh)
			else:
				host = t.empty(words, dtype=t.int64, pin_memory=True)
				host.copy_(pack, non_blocking=True)
				t.cuda.current_stream(self.device).synchronize()
				flat = host.numpy()
		if os.environ.get('NWAY_DOWNLOAD', '') == 'copy':
			flat = flat.copy()  # (pageable memory of the caller's own)
		idx = [flat[c * n:(c + 1) * n] for c in range(ni)]
		small = (flat[ni * n:(ni + 1) * n], flat[(ni + 1) * n:(ni + 2) * n]) if ns else None
		f64 = [flat[(ni + ns + c) * n:(ni + ns + c + 1) * n].view(numpy.float64) for c in range(nf)]
		return idx, small, f64

	def close(self):
		if getattr(self, 'handle', None):
			self.lib.nwayhip_plan_destroy(self.handle)
			self.handle = None

	def release(self):
		"""the caller is done with the table: the plan (workspace, table, staging) stays for the next match of the same shape
		(plan cache below), or is closed.  Its columns are no longer the caller's: the next match of that shape overwrites them, so
		a MatchResult of a released plan refuses to hand them out (``check_live``).  The cache holds at most PLAN_CACHE_ENTRIES plans
		and PLAN_CACHE_BYTES of device memory that torch.cuda.empty_cache() cannot return: ``plan_cache_clear()`` frees it,
		NWAY_PLAN_CACHE=0 in the environment turns the cache off.  Plans run on the stream that is current when they are
		enqueued; a cached plan handed to another stream is ordered after its previous use only through the synchronisation
		that read its status (every run_plan ends in one)."""
		self.released = True
		_plan_cache_put(self)

	def check_live(self):
		if getattr(self, 'released', False) or getattr(self, 'handle', None) is None:
			raise NwayHipError('this match table was released (or its plan closed): its device columns belong to the next match of the same shape')

	def __del__(self):
		try:
			self.close()
		except Exception:
			pass


class ZoneBatch(object):
	"""Several settled plans -- the declination zones of one job -- enqueued as ONE launch set (include/nwayhip.h:
	nwayhip_zones_*): one registration, one sweep and one tail launch for all of them instead of three per zone.  Every plan
	keeps its workspace, table and status block; ``enqueue`` takes the zones' catalogue lists in the plans' order.  Where
	the plans do not qualify (see the header) the zones go out one after the other: ``batched`` says which it was."""

	REGISTRATION = dict(auto=0, atomics=1, owner=2)  # NWAYHIP_ZONES_REG_*

	def __init__(self, plans, registration='auto'):
		"""registration: how the set registers its primaries -- 'atomics' (every claim an atomic in memory, as a plan of its own),
		'owner' (owner-computes: records, buckets, one workgroup per slice of a table claiming in LDS; csrc/zones.inc), 'auto'
		(owner from 200 000 primaries per set on).  Same tables' invariants, same results."""
		self.lib = load()
		self.plans = list(plans)
		if not 1 <= len(self.plans) <= MAXZONES:
			raise ValueError('a launch set takes 1..%d zones' % MAXZONES)
		self.device = self.plans[0].device
		handles = (ctypes.c_void_p * len(self.plans))(*[p.handle for p in self.plans])
		h = ctypes.c_void_p()
		check(self.lib.nwayhip_zones_create(ctypes.byref(h), handles, len(self.plans)))
		self.handle = h
		check(self.lib.nwayhip_zones_set_registration(self.handle, self.REGISTRATION[registration]))
		t = torch()
		self.args_bytes = int(self.lib.nwayhip_zones_args_bytes(self.handle))
		self.args = t.empty(self.args_bytes + 256, dtype=t.uint8, device=self.device)
		self.args_ptr = (self.args.data_ptr() + 255) // 256 * 256
		self._runs = (ZoneRun * len(self.plans))()
		self._cats = [None] * len(self.plans)

	def enqueue(self, catalogues, stream=None):
		"""catalogues: per zone the list of its DeviceCatalogues; does not synchronise"""
		for z, (plan, cats) in enumerate(zip(self.plans, catalogues)):
			self._cats[z] = (Catalogue * plan.ncat)(*[c.struct() for c in cats])  # (kept alive until the call returns)
			r = self._runs[z]
			r.h_cats = self._cats[z]
			r.workspace = plan.ws_ptr
			r.workspace_bytes = plan.ws_len
			r.h_table = ctypes.pointer(plan.table_struct)
			r.d_status = plan.status.data_ptr()
		s = stream if stream is not None else current_stream_ptr(self.device)
		check(self.lib.nwayhip_zones_enqueue(self.handle, self._runs, ctypes.c_void_p(self.args_ptr), self.args_bytes, s))

	@property
	def batched(self):
		"""whether the last enqueue went out as one launch set"""
		return bool(self.lib.nwayhip_zones_batched(self.handle))

	@property
	def owner_computes(self):
		"""whether the last enqueue's registration was owner-computes"""
		return int(self.lib.nwayhip_zones_batched(self.handle)) == 2

	def close(self):
		if getattr(self, 'handle', None):
			self.lib.nwayhip_zones_destroy(self.handle)
			self.handle = None

	def __del__(self):
		try:
			self.close()
		except Exception:
			pass


def make_params(ncat, scheme, radius_arcsec, err_deg, dens, dens_plus, prior_table, prob_ratio_secondary=0.5,
		radius_filter=True, correction=CORRECTION_NONE, finalize=True, sphere_cell_factor=0.0, bitmap_bits=0, link_slots=0, table_slots=0, f32_roundtrip=False,
		tuning=None):
	"""tuning: dict with any of direct_log2, fold_log2, disable (DISABLE_* mask), enable (ENABLE_* mask), link_slots, sphere_cell_factor -- what tests and
	benchmarks use to force a path (nwayhip.h: nwayhip_match_params); None = the library decides"""
	p = MatchParams()
	tuning = dict(tuning or {})
	link_slots = tuning.pop('link_slots', link_slots)
	p.direct_log2 = int(tuning.pop('direct_log2', 0))
	p.fold_log2 = int(tuning.pop('fold_log2', 0))
	p.disable = int(tuning.pop('disable', 0))
	p.enable = int(tuning.pop('enable', 0))
	sphere_cell_factor = float(tuning.pop('sphere_cell_factor', sphere_cell_factor))
	if tuning:
		raise ValueError('unknown tuning keys: %s' % sorted(tuning))
	p.table_slots = table_slots
	p.f32_roundtrip = 1 if f32_roundtrip else 0
	p.link_slots = link_slots
	p.ncat = ncat
	p.scheme = scheme
	p.radius_filter = 1 if radius_filter else 0
	p.correction = correction
	p.finalize = 1 if finalize else 0
	p.err_deg = err_deg
	p.radius_arcsec = radius_arcsec
	p.prob_ratio_secondary = prob_ratio_secondary
	for c in range(ncat):
		p.dens[c] = dens[c]
		p.dens_plus[c] = dens_plus[c]
	for i, v in enumerate(prior_table):
		p.prior_table[i] = v
	p.sphere_cell_factor = sphere_cell_factor
	p.bitmap_bits = bitmap_bits
	return p


CAPACITY_LIMIT = (1 << 31) - 4096  # row / link positions are int32 on the device


# Plans of recent matches, kept for the next one of the same shape: nway_match on catalogues of a few hundred thousand rows
# is host time (plan creation, 20-odd allocations, the capacity estimate and its repeats), not GPU time.  A released plan
# waits here with its buffers; the key is everything it was created from.  `settled`: what a request ended up with after
# its overflow repeats, so that the next identical request starts there.
PLAN_CACHE_ENTRIES = 4
PLAN_CACHE_BYTES = 1 << 30
_plan_cache = []      # [(key, plan)], most recent last
_plan_settled = {}    # request key -> (params bytes, cap_pairs, cap_rows)
_plan_lock = threading.Lock()


def _plan_key(sizes, params, cap_pairs, cap_rows, device, lean):
	return (tuple(int(n) for n in sizes), bytes(params), int(cap_pairs), int(cap_rows), str(device), bool(lean))


def _plan_cache_get(key):
	with _plan_lock:
		for i, (k, plan) in enumerate(_plan_cache):
			if k == key:
				del _plan_cache[i]
				plan.released = False
				return plan
	return None


def _plan_cache_put(plan):
	if getattr(plan, 'handle', None) is None:
		return
	key = getattr(plan, 'cache_key', None)
	if key is None or plan.device_bytes > PLAN_CACHE_BYTES or os.environ.get('NWAY_PLAN_CACHE', '1') == '0':
		plan.close()
		return
	evict = []
	with _plan_lock:
		_plan_cache.append((key, plan))
		while len(_plan_cache) > PLAN_CACHE_ENTRIES or sum(p.device_bytes for _, p in _plan_cache) > PLAN_CACHE_BYTES:
			evict.append(_plan_cache.pop(0)[1])
	for p in evict:
		p.close()


def plan_cache_clear():
	with _plan_lock:
		old = [p for _, p in _plan_cache]
		del _plan_cache[:]
		_plan_settled.clear()
	for p in old:
		p.close()


def _new_plan(sizes, params, cap_pairs, cap_rows, device, lean):
	key = _plan_key(sizes, params, cap_pairs, cap_rows, device, lean)
	plan = _plan_cache_get(key)
	if plan is None:
		plan = MatchPlan(sizes, params, cap_pairs, cap_rows, device, lean=lean)
		plan.cache_key = key
	return plan


def run_plan(sizes, params, catalogues, cap_pairs, cap_rows, device, max_retries=6, lean=False):
	"""enqueue, synchronise, grow the capacities on overflow; returns (plan, status).

	Every kind of overflow has its own budget of ``max_retries`` repeats (a clustered k >= 4 match
	may need the general path, a larger link region AND several row doublings): the status words
	report exact needs where a run could count them (links, the rows of a 2-way table), a lower
	bound otherwise (an expansion level that overflowed ends the run: rows grow fourfold then)."""
	tries = dict(table=0, path=0, pairs=0, rows=0, slots=0)
	attempts = 0
	request = _plan_key(sizes, params, cap_pairs, cap_rows, device, lean)
	params = type(params).from_buffer_copy(params)  # the caller's struct is left as it was handed over; plan.params = what was settled on
	with _plan_lock:
		known = _plan_settled.get(request)
	if known is not None:  # what this very request ended up with the last time
		params = type(params).from_buffer_copy(known[0])
		cap_pairs, cap_rows = known[1], known[2]
	while True:
		plan = _new_plan(sizes, params, cap_pairs, cap_rows, device, lean)
		plan.enqueue(catalogues)
		attempts += 1
		plan.attempts = attempts  # enqueues it took to settle the capacities (1 = the first guess held)
		st = plan.read_status()
		flags = int(st[ST_FLAGS])
		if os.environ.get('NWAYHIP_TRACE'):
			sys.stderr.write('run_plan: link_slots %d flags %d rows %d cap_pairs %d cap_rows %d\n' % (plan.link_slots, flags, int(st[ST_ROWS]), cap_pairs, cap_rows))
		if flags == 0:
			if attempts > 1:
				with _plan_lock:
					if len(_plan_settled) < 64:
						_plan_settled[request] = (bytes(params), int(cap_pairs), int(cap_rows))
			return plan, st
		sparse = plan.sparse
		fused = plan.fused
		slots = plan.table_slots()
		plan.close()
		del plan

		def spend(kind):
			tries[kind] += 1
			if tries[kind] > max_retries:
				raise NwayHipError('match table capacity could not be settled (%s, %d attempts; status flags %d)' % (kind, max_retries, flags))
		if flags & FLAG_QUAD_DEEP:
			# a primary with three or more candidates in one catalogue: the 3-way tail with four lanes per primary
			# leaves those to the one that walks them (csrc/tail3q.inc)
			params.disable = int(params.disable) | DISABLE_QUAD3
			flags &= ~FLAG_QUAD_DEEP
			if flags == 0 or flags == FLAG_ROW_OVERFLOW:  # (the rows of those primaries were not counted)
				spend('path')
				continue
		need = int(st[ST_SLOT_NEED])
		if flags & FLAG_SLOT_OVERFLOW and not flags & (FLAG_LOOKBACK | FLAG_REG_OVERFLOW) and 0 < need <= LINK_SLOTS_CAP and tries['slots'] < 2:
			# a primary with more candidates than the slots sized for the mean density (a clustered
			# field): the run counted them; once or twice more with that many slots before giving the sparse front up
			tries['slots'] += 1
			params.link_slots = min(LINK_SLOTS_CAP, need + (need >> 3) + 1)
			continue
		if flags & (FLAG_SLOT_OVERFLOW | FLAG_LOOKBACK) or (flags & FLAG_REG_OVERFLOW and sparse):
			# the sparse path does not fit this input (a primary with more candidates than slots,
			# primaries piled up in a few cells): repeat on the general path
			spend('path')
			params.link_slots = -1
			continue
		if flags & FLAG_REG_OVERFLOW:
			# the cell table is sized for the expected registrations per primary; catalogues piled
			# up near a pole need more: come back with a larger table (bounded)
			spend('table')
			slots *= 4
			if slots > 256 * max(int(sizes[0]), 1) + (1 << 16):
				raise NwayHipError('primary cell registration overflowed (sources piled up on a pole?)')
			params.table_slots = slots
			continue
		if flags & FLAG_PAIR_OVERFLOW:
			spend('pairs')
			need_pairs = int(max(st[ST_PAIRS:ST_PAIRS + 8]))
			cap_pairs = max(cap_pairs, int(need_pairs * 1.05) + 1024)
			if int(st[ST_REGION_NEED]) > 0:
				# the total may fit, but one workgroup's region of the link arrays did not (clustered input)
				params.link_region_min = max(int(params.link_region_min), int(int(st[ST_REGION_NEED]) * 1.3) + 64)
		if flags & FLAG_ROW_OVERFLOW:
			spend('rows')
			if cap_rows >= CAPACITY_LIMIT:
				raise NwayHipError('the match table would have more than 2^31 rows: split the primary catalogue')
			reported = int(st[ST_ROWS])
			if flags & FLAG_PAIR_OVERFLOW:
				cap_rows = cap_rows * 2             # the links were incomplete: the row count means little
			elif params.ncat == 2 or fused:
				cap_rows = int(reported * 1.02) + 1024  # exact (the fused tails count every row they cannot write): one repeat settles it
			else:
				cap_rows = max(cap_rows * 4, int(reported * 1.5) + 1024)  # a level overflowed: lower bound only
			cap_rows = min(cap_rows, CAPACITY_LIMIT)
		if cap_pairs > CAPACITY_LIMIT:
			raise NwayHipError('more than 2^31 links: split the primary catalogue')
		if os.environ.get('NWAY_NO_EMPTY_CACHE', '') != '1':  # (development: tools/dev/fault_study.sh tells the variants apart)
			torch().cuda.empty_cache()
