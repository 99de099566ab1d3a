"""Multi-GPU sharding of the match: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" for the CPU tests).

The unit of work is the primary source: every output row belongs to exactly one primary
and p_any / p_i / match_flag reduce only within a primary's rows (__init__.py:459,
nway.py:547), so the primary catalogue is sharded by contiguous row ranges and each rank
produces the corresponding contiguous block of the global match table.  Priors depend on
the secondary densities only (nu_0 cancels against nu+_0 = nu_0, __init__.py:214,254), so
a shard's rows are bit-identical to the same rows of the unsharded run.

The one exchange step is the distribution of the secondary catalogues: each rank loads a
slice and an all-gatherv leaves the full ra / dec / error columns resident on every GPU --
through torch.distributed as ONE RCCL all-gather of equally padded pieces (the default), or,
with ``comm='rccl'``, through the library's own RCCL calls behind the C ABI as one group of
broadcasts, no padding (csrc/comm.inc: nwayhip_comm_allgatherv_f64).  It happens once per catalogue, not
once per primary batch; ``ShardedMatch.setup`` times it separately.  No collective is on
the per-batch path; the host concatenates per-rank tables in rank order when a global
table is wanted.
"""
from __future__ import division, print_function

import time

import numpy


def _dist():
	import torch.distributed as dist
	return dist


_side_streams = {}


def side_streams(device, n):
	"""``n`` HIP streams beside the current one, made once per device and handed out again: the runtime multiplexes streams onto a
	handful of hardware queues (four by default), and every further stream a process creates may land on the queue of one it is
	meant to run beside -- measured: the eight zones of a 5e5 x 1e8 job on two streams took 543 us per pass in a fresh process and
	684 us behind two other streams made earlier (bench.py's two_pipelines leg) -- so everything in this package that wants side
	streams takes them from here"""
	import torch
	key = str(torch.device(device))
	have = _side_streams.setdefault(key, [])
	while len(have) < n:
		have.append(torch.cuda.Stream(device=device))
	return have[:n]


def ctypes_stream(stream):
	"""a torch stream as the ``void*`` the C ABI takes"""
	import ctypes
	return ctypes.c_void_p(stream.cuda_stream)


def world_info(group=None):
	dist = _dist()
	if dist.is_available() and dist.is_initialized():
		return dist.get_rank(group), dist.get_world_size(group)
	return 0, 1


def make_comm(device, group=None):
	"""an RCCL communicator behind the C ABI (``_hip.RcclComm``) over the ranks of ``group``: rank 0 makes the id, one
	broadcast over the process group that launched the ranks hands it out"""
	from nway_amd import _hip
	dist = _dist()
	rank, world = world_info(group)
	ident = [_hip.RcclComm.unique_id() if rank == 0 else None]
	if world > 1:
		src = 0 if group is None else dist.get_global_rank(group, 0)
		dist.broadcast_object_list(ident, src=src, group=group)
	return _hip.RcclComm(world, rank, ident[0], device)


def allgatherv(tensor, group=None, comm=None):
	"""Concatenation over ranks (rank order) of 1-D tensors of different lengths.
	Returns (full tensor, list of per-rank counts).

	comm (an ``_hip.RcclComm``): the column travels through the library's own RCCL calls (nwayhip_comm_allgatherv_f64:
	one group of broadcasts); only the row counts go through torch.distributed.

	RCCL ("nccl"): ONE all-gather of equal pieces -- every rank pads its slice to the longest (the slices of a
	catalogue differ by a row or two; RCCL has no native all-gatherv and the pieces of a ring all-gather must be
	equal), the padding is dropped when the pieces are copied into place.  gloo (CPU tests, ranks sharing a GPU):
	one broadcast per rank."""
	import torch
	dist = _dist()
	rank, world = world_info(group)
	if world == 1:
		n = int(tensor.shape[0])
		return (tensor if comm is None else comm.allgatherv(tensor, [n])), [n]
	count = torch.tensor([tensor.shape[0]], dtype=torch.int64, device=tensor.device)
	counts = [torch.zeros(1, dtype=torch.int64, device=tensor.device) for _ in range(world)]
	dist.all_gather(counts, count, group=group)
	counts = [int(c.item()) for c in counts]
	if comm is not None:
		return comm.allgatherv(tensor, counts), counts
	full = torch.empty(sum(counts), dtype=tensor.dtype, device=tensor.device)
	offsets = numpy.concatenate([[0], numpy.cumsum(counts)])
	pieces = [full[offsets[r]:offsets[r + 1]] for r in range(world)]
	# (the padded all-gather is also what gloo runs on host tensors: the CPU tests walk the very lines the RCCL run takes;
	# gloo with device tensors -- ranks sharing one GPU -- keeps to broadcasts, which it is known to carry)
	if dist.get_backend(group) == 'nccl' or tensor.device.type == 'cpu':
		longest = max(counts)
		if longest == 0:
			return full, counts
		if min(counts) == longest:
			dist.all_gather_into_tensor(full, tensor.contiguous(), group=group)
		else:
			mine = torch.zeros(longest, dtype=tensor.dtype, device=tensor.device)
			mine[:counts[rank]] = tensor
			padded = torch.empty(world * longest, dtype=tensor.dtype, device=tensor.device)
			dist.all_gather_into_tensor(padded, mine, group=group)
			for r in range(world):
				pieces[r].copy_(padded[r * longest:r * longest + counts[r]])
	else:
		pieces[rank].copy_(tensor)
		works = []
		for r in range(world):
			if counts[r] > 0:
				src = r if group is None else dist.get_global_rank(group, r)
				works.append(dist.broadcast(pieces[r], src=src, group=group, async_op=True))
		for w in works:
			w.wait()
	return full, counts


def shard_bounds(n, world):
	"""contiguous, balanced row ranges: rank r owns [bounds[r], bounds[r+1])"""
	base, extra = divmod(n, world)
	sizes = [base + (1 if r < extra else 0) for r in range(world)]
	return numpy.concatenate([[0], numpy.cumsum(sizes)]).astype(numpy.int64)



class MagnitudePriors(object):
	"""Magnitude (or other property) priors on a SHARDED match (nwaylib/__init__.py:304-396, ``_apply_magnitude_biasing`` and the
	final probabilities :399-461) -- a mix-in of ``ShardedMatch`` and ``ZoneShardedMatch``.

	What the reference computes on the one table splits into a GLOBAL part -- which sources of a catalogue are secure counterparts
	and which are field sources decides the two histograms of a column (:324-375) -- and a per-row part (the bias of a row and the
	statistics of its primary's group, :383-394, :399-461).  The global part needs, of every rank's rows that HAVE a counterpart in
	the column's catalogue, four small things: the counterpart's index, whether the row is secure / plausible, its weight.  They are
	gathered (``all_gather_object``; the table itself stays where it is), put into the order of the global table (stable by primary:
	the reference looks a secure source's weight up at the position of its first secure row, :337 -- the order matters), and every
	rank then evaluates the SAME selection and histograms with ``magpriors.secure_and_field_sources`` (the code ``nway_match`` itself
	runs on one GPU).  The per-row part runs on the rank's own rows: ``nwayhip_bias_lookup`` and ``nwayhip_group_stats`` on the device.

	Hooks (the CPU tests fill them with numpy): ``_bias_lookup``, ``_final_probabilities``."""

	def _allgather_column(self, values):
		"""a column given as this rank's shard / slice (contiguous global rows, rank order) -> the whole column on every rank"""
		import torch
		col, _ = allgatherv(torch.as_tensor(numpy.asarray(values, dtype=float)).to(self._exchange_device()), self.group, None)
		return col.cpu().numpy()

	def _bias_lookup(self, idx, mag_all, func, total):
		"""log10(func(mag_all[idx])) added to ``total`` in place (undefined -> 0); returns the bias column 10^weight (device kernel)"""
		from nway_amd import _hip
		lib = _hip.load()
		t = _hip.torch()
		n = len(idx)
		if n == 0:
			return numpy.zeros(0)
		d_idx = _hip.to_device(numpy.asarray(idx, dtype=numpy.int32), self.device, dtype=t.int32)
		d_mag = _hip.to_device(numpy.where(numpy.isfinite(mag_all), mag_all, numpy.nan), self.device)
		d_edges = _hip.to_device(func.edges, self.device)
		d_ratio = _hip.to_device(func.values, self.device)
		d_total = _hip.to_device(total, self.device)
		d_bias = t.empty(n, dtype=t.float64, device=self.device)
		_hip.check(lib.nwayhip_bias_lookup(n, _hip.ptr(d_idx), _hip.ptr(d_mag), len(func.edges), _hip.ptr(d_edges), _hip.ptr(d_ratio),
			_hip.ptr(d_total), _hip.ptr(d_bias), _hip.current_stream_ptr(self.device)))
		total[:] = _hip.to_host(d_total)
		return _hip.to_host(d_bias)

	def _final_probabilities(self, primary_index, ncat, total, prior, ratio):
		"""p_single, p_any, p_i, match_flag of this rank's rows from the biased totals (device kernel); rows of a primary are contiguous"""
		from nway_amd import _hip
		lib = _hip.load()
		t = _hip.torch()
		n = len(total)
		if n == 0:
			z = numpy.zeros(0)
			return z, z, z, numpy.zeros(0, dtype=numpy.int64)
		starts = numpy.flatnonzero(numpy.r_[True, primary_index[1:] != primary_index[:-1]])
		group_start = numpy.r_[starts, n].astype(numpy.int64)
		d_gs = _hip.to_device(group_start, self.device, dtype=t.int64)
		d_total = _hip.to_device(total, self.device)
		d_prior = _hip.to_device(prior, self.device)
		out = [t.empty(n, dtype=t.float64, device=self.device) for _ in range(3)]
		flag = t.empty(n, dtype=t.int8, device=self.device)
		_hip.check(lib.nwayhip_group_stats(n, len(starts), _hip.ptr(d_gs), _hip.ptr(d_total), _hip.ptr(d_prior), float(ratio), _hip.ptr(out[0]),
			_hip.ptr(out[1]), _hip.ptr(out[2]), _hip.ptr(flag), _hip.current_stream_ptr(self.device)))
		return _hip.to_host(out[0]), _hip.to_host(out[1]), _hip.to_host(out[2]), _hip.to_host(flag).astype(numpy.int64)

	def magnitude_priors(self, mags, mag_include_radius=None, mag_exclude_radius=None, magauto_post_single_minvalue=0.9,
			store_mag_hists=False, logger=None):
		"""COLLECTIVE (every rank calls it, after ``step()``).  mags: per catalogue, primary first, a list of
		``(magname, values, maghist)`` -- ``values`` = the column for THIS RANK's shard / slice of the catalogue (as its ra / dec were
		given), ``maghist`` = None (learn the histogram from the match, "auto") or ``(bins_lo, bins_hi, hist_sel, hist_all)``.
		Returns this rank's block of the table with the ``bias_*`` columns and p_single / match_flag / prob_has_match /
		prob_this_match recomputed from the biased totals -- row for row what ``nway_amd.nway_match`` returns on one GPU."""
		import nway_amd
		from nway_amd import magnitudeweights, magpriors
		logger = logger or nway_amd.NullOutputLogger()
		if mag_exclude_radius is None:
			mag_exclude_radius = mag_include_radius
		t = self.local_table()
		names = [self.primary['name']] + [s['name'] for s in self.secondary_slices]
		k = len(names)
		prim = numpy.asarray(t[names[0]], dtype=numpy.int64)
		n = len(prim)
		total = numpy.array(t['dist_bayesfactor'], dtype=float) if n else numpy.zeros(0)
		dist = _dist()
		for c, columns in enumerate(mags):
			for magname, values, maghist in columns:
				col = '%s_%s' % (names[c], magname)
				mag = '%s:%s' % (names[c], magname)
				mag_all = self._allgather_column(values) if self.world > 1 else numpy.array(values, dtype=float)
				mag_all[mag_all == -99] = numpy.nan
				idx = numpy.asarray(t[names[c]], dtype=numpy.int64)
				if maghist is None:
					if mag_include_radius is not None:
						sep_max = numpy.asarray(t['Separation_max'])
						secure, plausible, weights = sep_max < mag_include_radius, sep_max < mag_exclude_radius, numpy.ones(n)
					else:
						post = numpy.asarray(t['dist_post'])
						secure, plausible, weights = post > magauto_post_single_minvalue, post > 0.01, post
					present = idx != -1
					mine = (prim[present], idx[present], numpy.asarray(secure)[present], numpy.asarray(plausible)[present], numpy.asarray(weights, dtype=float)[present])
					if self.world > 1:
						parts = [None] * self.world
						dist.all_gather_object(parts, mine, group=self.group)
					else:
						parts = [mine]
					g = [numpy.concatenate([numpy.asarray(part[i]) for part in parts]) for i in range(5)]
					order = numpy.argsort(g[0], kind='stable')  # the global table's order: by primary, a primary's rows as its rank has them
					target, target_weights, field, n_plausible = magpriors.secure_and_field_sources(g[1][order], mag_all, g[2][order].astype(bool),
						g[3][order].astype(bool), g[4][order], 'api', mag)
					logger.log('magnitude histogram of column "%s": %d secure matches, %d insecure matches and %d secure non-matches of %d total entries (%d valid)'
						% (col, len(target), n_plausible, field.sum(), len(mag_all), numpy.isfinite(mag_all).sum()))
					bins, hist_sel, hist_all = magnitudeweights.adaptive_histograms(mag_all[field], target, weights=target_weights)
					if store_mag_hists and self.rank == 0:
						magpriors.write_histogram(mag.replace(':', '_') + '_fit.txt', bins, hist_sel, hist_all)
					if len(target) < 100:
						raise nway_amd.UndersampledException('ERROR: too few secure matches (%d) to make a good histogram. If you are sure you want to use this poorly sampled histogram, replace "auto" with the filename. You can also decrease the mag-auto-minprob parameter.' % len(target))
				else:
					bins_lo, bins_hi, hist_sel, hist_all = maghist
					bins = numpy.array(list(bins_lo) + [bins_hi[-1]])
				func = magnitudeweights.fitfunc_histogram(bins, hist_sel, hist_all)
				t['bias_%s' % col] = self._bias_lookup(idx, mag_all, func, total)
		# the prior of a row follows from which catalogues it has (__init__.py:254), with the densities of the WHOLE catalogues
		comp = nway_amd._completeness_vector(self.prior_completeness, k)
		table = nway_amd._prior_table(numpy.asarray(self.dens), numpy.asarray(self.dens_plus), comp)
		pattern = numpy.zeros(n, dtype=numpy.int64)
		for c in range(1, k):
			pattern |= (numpy.asarray(t[names[c]]) >= 0).astype(numpy.int64) << (c - 1)
		prior = table[pattern]
		p_single, p_any, p_i, flag = self._final_probabilities(prim, numpy.asarray(t['ncat']), total, prior, self.prob_ratio_secondary)
		t['p_single'], t['prob_has_match'], t['prob_this_match'], t['match_flag'] = p_single, p_any, p_i, flag
		return t

	def gather_magnitude_table(self, local, dst=0):
		"""the global table of ``magnitude_priors`` on rank ``dst`` (sorted by primary), None elsewhere"""
		if self.world == 1:
			return local
		gathered = [None] * self.world if self.rank == dst else None
		_dist().gather_object(local, gathered, dst=dst, group=self.group)
		if self.rank != dst:
			return None
		out = dict((key, numpy.concatenate([numpy.asarray(g[key]) for g in gathered])) for key in gathered[0] if not key.startswith('_'))
		order = numpy.argsort(out[self.primary['name']], kind='stable')
		return dict((key, v[order]) for key, v in out.items())


class ShardedMatch(MagnitudePriors):
	"""Primary rows sharded over the ranks, secondary catalogues replicated by all-gatherv.

	primary: this rank's shard of the primary catalogue (dict: name, ra, dec, error, area)
	secondaries: list of this rank's SLICES of the secondary catalogues (same dict layout;
	  ``error`` may be a scalar)
	tuning: development / test knobs of the plan (``_hip.make_params``), normally None

	The exchange logic (``setup``, ``total_rows``, ``gather_table``) only touches the hooks
	``_exchange_device``, ``_sync``, ``_build_plan``, ``step``, ``local_rows``, ``local_table``;
	here they are the HIP pipeline and nothing else (no CPU path in this package).  The CPU
	tests (``tests/test_distributed_gloo.py``) subclass and fill the hooks with the oracle.
	"""

	def __init__(self, primary, secondaries, match_radius, prior_completeness, device, group=None,
			prob_ratio_secondary=0.5, tuning=None, comm=None):
		if isinstance(secondaries, dict):
			secondaries = [secondaries]
		self.comm = make_comm(device, group) if comm == 'rccl' else comm  # None: torch.distributed moves the columns
		self.primary = primary
		self.secondary_slices = secondaries
		self.match_radius = float(match_radius)
		self.prior_completeness = prior_completeness
		self.prob_ratio_secondary = prob_ratio_secondary
		self.device = device
		self.group = group
		self.tuning = tuning
		self.rank, self.world = world_info(group)
		self.plan = None
		self.setup_seconds = None
		self.setup()

	# -- hooks ---------------------------------------------------------------------------
	def _exchange_device(self):
		"""where the exchanged columns live (the GPU: RCCL moves them over xGMI)"""
		return self.device

	def _sync(self):
		import torch
		torch.cuda.synchronize(self.device)

	# -- one-time exchange ---------------------------------------------------------------
	def setup(self):
		import torch
		dist = _dist()
		t0 = time.perf_counter()
		dev = self._exchange_device()
		self.full_secondaries = []
		self.gathered_bytes = 0
		for sl in self.secondary_slices:
			ra, _ = allgatherv(torch.as_tensor(numpy.asarray(sl['ra'], dtype=float)).to(dev), self.group, self.comm)
			dec, counts = allgatherv(torch.as_tensor(numpy.asarray(sl['dec'], dtype=float)).to(dev), self.group, self.comm)
			self.gathered_bytes += 16 * int(ra.shape[0])
			if numpy.ndim(sl['error']) == 0:
				err = float(sl['error'])
			else:
				err, _ = allgatherv(torch.as_tensor(numpy.asarray(sl['error'], dtype=float)).to(dev), self.group, self.comm)
				self.gathered_bytes += 8 * int(ra.shape[0])
			self.full_secondaries.append(dict(name=sl['name'], ra=ra, dec=dec, error=err, area=sl['area'], counts=counts))
		# global size of the primary catalogue (only logged; the priors do not depend on it)
		n0 = torch.tensor([len(self.primary['ra'])], dtype=torch.int64, device=dev)
		if self.world > 1:
			sizes = [torch.zeros_like(n0) for _ in range(self.world)]
			dist.all_gather(sizes, n0, group=self.group)
			self.primary_sizes = [int(s.item()) for s in sizes]
		else:
			self.primary_sizes = [int(n0.item())]
		self.primary_offset = int(sum(self.primary_sizes[:self.rank]))
		self._sync()
		self.setup_seconds = time.perf_counter() - t0
		import nway_amd
		self.dens, self.dens_plus = nway_amd._densities_from_sizes([self.primary['name']] + [f['name'] for f in self.full_secondaries],
			[int(sum(self.primary_sizes))] + [int(f['ra'].shape[0]) for f in self.full_secondaries],
			[self.primary['area']] + [f['area'] for f in self.full_secondaries], nway_amd.NullOutputLogger())
		self._build_plan()

	def _tables(self):
		"""match_tables of this rank: own primary shard + complete secondaries"""
		sec = []
		for f in self.full_secondaries:
			e = f['error']
			sec.append(dict(name=f['name'], ra=f['ra'], dec=f['dec'], error=e, area=f['area'], mags=[], maghists=[], magnames=[]))
		prim = dict(self.primary)
		# density of the WHOLE primary catalogue: scale the area of the shard (the CPU stand-ins of the
		# tests derive the densities from the tables they are given)
		total = float(sum(self.primary_sizes))
		if total > 0 and len(self.primary['ra']) > 0:
			prim['area'] = self.primary['area'] * (len(self.primary['ra']) / total)
		return [prim] + sec

	def _build_plan(self):
		import nway_amd
		from nway_amd import _hip
		tables = self._tables()
		log = nway_amd.NullOutputLogger()
		self.empty = len(self.primary['ra']) == 0  # (a rank without primaries still takes part in every collective)
		k = len(tables)
		err = self.match_radius / 60. / 60
		# the flat-vs-all-sky decision needs every catalogue's extent: primaries are sharded, so
		# the decision is made per rank and then agreed on (all-sky wins)
		scheme = nway_amd.choose_scheme([(numpy.asarray(self.primary['ra'], dtype=float), numpy.asarray(self.primary['dec'], dtype=float))], err)
		if scheme == _hip.SCHEME_FLAT:
			scheme = _hip.scheme_from_extents([_hip.catalogue_extent(t['ra'], t['dec']) for t in tables[1:]], err)
		if self.world > 1:
			import torch
			s = torch.tensor([scheme], dtype=torch.int64, device=self.device)
			_dist().all_reduce(s, op=_dist().ReduceOp.MAX, group=self.group)
			scheme = int(s.item())
		self.scheme = scheme
		# densities of the WHOLE catalogues (the global primary count over the full area: the same nu_0
		# on every rank, also on one whose shard is empty)
		dens, dens_plus = nway_amd._densities_from_sizes([t['name'] for t in tables],
			[int(sum(self.primary_sizes))] + [int(t['ra'].shape[0]) for t in tables[1:]], [self.primary['area']] + [t['area'] for t in tables[1:]], log)
		comp = nway_amd._completeness_vector(self.prior_completeness, k)
		self.params = _hip.make_params(k, scheme, self.match_radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp),
			prob_ratio_secondary=self.prob_ratio_secondary, tuning=self.tuning)
		self.cats = [_hip.DeviceCatalogue(self.primary['ra'], self.primary['dec'], numpy.asarray(self.primary['error'], dtype=float), self.device)]
		for t in tables[1:]:
			self.cats.append(_hip.DeviceCatalogue(t['ra'], t['dec'], t['error'], self.device))
		sizes = [c.n for c in self.cats]
		cap_pairs, cap_rows = nway_amd._estimate_capacities(sizes, [t['area'] * 1.0 for t in tables], self.match_radius, scheme, True)
		if self.empty:
			self.plan, self.status = None, numpy.zeros(_hip.STATUS_WORDS, dtype=numpy.int64)
			return
		self.plan, self.status = _hip.run_plan(sizes, self.params, self.cats, cap_pairs, cap_rows, self.device, lean=True)
		self.params = self.plan.params  # (what the run settled on: slots, table size, path -- not the request)

	# -- per batch -----------------------------------------------------------------------
	def step(self, cats=None):
		"""one pass of the hot path over this rank's primary shard (no collective).  cats: other device copies of the same
		catalogues (bench.py alternates over copies of the secondaries so that no pass finds its stream in the Infinity Cache)"""
		if not self.empty:
			self.plan.enqueue(self.cats if cats is None else cats)

	def read_status(self):
		return self.status if self.plan is None else self.plan.read_status()

	def pass_bytes(self, rows):
		"""algorithmic bytes of this rank's pass: its primaries and every secondary catalogue
		read once (ra, dec, the positional error where it is a column) + the output columns of its rows"""
		k = 1 + len(self.full_secondaries)
		b = len(self.primary['ra']) * (16.0 + (8.0 if numpy.ndim(self.primary['error']) > 0 else 0.0))
		for f in self.full_secondaries:
			b += int(f['ra'].shape[0]) * (16.0 + (0.0 if numpy.ndim(f['error']) == 0 and not hasattr(f['error'], 'shape') else 8.0))
		per_row = 4 * k + 8 * (k * (k - 1) // 2) + 8 + 1 + 8 * 5 + 1
		return b + per_row * rows

	def local_rows(self):
		from nway_amd import _hip
		return int(self.read_status()[_hip.ST_ROWS])

	def total_rows(self):
		"""rows produced by all ranks in one step"""
		import torch
		n = self.local_rows()
		if self.world == 1:
			return n
		t = torch.tensor([n], dtype=torch.int64, device=self._exchange_device())
		_dist().all_reduce(t, group=self.group)
		return int(t.item())

	def local_table(self):
		"""this rank's block of the global table as host columns (global primary indices)"""
		if self.plan is None:
			names = [self.primary['name']] + [f['name'] for f in self.full_secondaries]
			from nway_amd import _hip
			t = dict((nme, numpy.zeros(0, dtype=numpy.int64)) for nme in names + ['ncat', 'match_flag'])
			for i, j in _hip.pair_columns(len(names)):
				t['Separation_%s_%s' % (names[i], names[j])] = numpy.zeros(0)
			for dst in ('Separation_max', 'dist_bayesfactor_uncorrected', 'dist_bayesfactor', 'dist_post', 'p_single', 'prob_has_match', 'prob_this_match'):
				t[dst] = numpy.zeros(0)
		else:
			from nway_amd import _hip
			st = self.plan.read_status()
			m = int(st[_hip.ST_ROWS])
			names = [self.primary['name']] + [f['name'] for f in self.full_secondaries]
			t = {}
			for c, nme in enumerate(names):
				t[nme] = _hip.to_host(self.plan.cols['idx'][c][:m]).astype(numpy.int64)
			for p, (i, j) in enumerate(_hip.pair_columns(len(names))):
				t['Separation_%s_%s' % (names[i], names[j])] = _hip.to_host(self.plan.cols['sep'][p][:m])
			for src, dst in (('sep_max', 'Separation_max'), ('log_bf', 'dist_bayesfactor_uncorrected'), ('log_bf_corrected', 'dist_bayesfactor'),
					('dist_post', 'dist_post'), ('p_single', 'p_single'), ('p_any', 'prob_has_match'), ('p_i', 'prob_this_match')):
				t[dst] = _hip.to_host(self.plan.cols[src][:m])
			t['ncat'] = _hip.to_host(self.plan.cols['ncat'][:m]).astype(numpy.int64)
			t['match_flag'] = _hip.to_host(self.plan.cols['match_flag'][:m]).astype(numpy.int64)
		pname = self.primary['name']
		t[pname] = numpy.asarray(t[pname]) + self.primary_offset
		return t

	def gather_table(self, dst=0):
		"""global table on rank ``dst`` (rank-order concatenation); None elsewhere"""
		dist = _dist()
		local = self.local_table()
		if self.world == 1:
			return local
		gathered = [None] * self.world if self.rank == dst else None
		dist.gather_object(local, gathered, dst=dst, group=self.group)
		if self.rank != dst:
			return None
		out = {}
		for key in gathered[0]:
			if key.startswith('_'):
				continue
			out[key] = numpy.concatenate([numpy.asarray(g[key]) for g in gathered])
		return out


class SecondarySplitMatch(object):
	"""ONE job over several GPUs (strong scaling; SURVEY.md 8(e), second half): the secondary STREAM
	is split.  Every rank registers ALL primaries (their cell table is small), sweeps only its own
	slice of every secondary catalogue, and exports each candidate (primary, secondary) to the rank
	that owns the primary -- contiguous row ranges, so the rank-order concatenation of the tables is
	the global table.  One all-to-all of fixed-size export buffers per step (RCCL over xGMI; a few
	hundred KB per peer) is the only collective; the owner turns what arrives into links and runs the
	fused tail on its own primaries.  Needs the sparse path with its one-launch fused tail: few chance neighbours per
	primary, AND at most four catalogues (three with the script's unrelated-association correction) -- five or more take the
	hybrid path (csrc/plan.inc: TAILK_MAX_K), which this mode does not drive.  Everything else shards by primary rows
	(``ShardedMatch``) or by declination zones (``ZoneShardedMatch``).

	primary: this rank's shard of the primary catalogue (the shards are all-gathered once at set-up)
	secondaries: list of this rank's SLICES of the secondary catalogues (``error`` may be a scalar)
	capacity: records per (destination, catalogue) block of the export buffer (None: a first guess from the sizes, then
	  what the settling step counted in the fullest block of any rank, doubled: the exchange ships whole blocks)
	tuning: development / test knobs of the plan (``_hip.make_params``), normally None

	As in ``ShardedMatch`` the exchange logic only touches the hooks ``_exchange_device``, ``_sync``,
	``_build_plan``, ``step``, ``local_rows``, ``local_table``: here the HIP pipeline, in the CPU tests
	(``tests/test_distributed_gloo.py``) a subclass around the oracle.
	"""

	def __init__(self, primary, secondaries, match_radius, prior_completeness, device, group=None,
			prob_ratio_secondary=0.5, capacity=None, tuning=None, comm=None):
		if isinstance(secondaries, dict):
			secondaries = [secondaries]
		self.comm = make_comm(device, group) if comm == 'rccl' else comm  # None: torch.distributed carries the exchanges
		self.primary = primary
		self.secondary_slices = secondaries
		self.match_radius = float(match_radius)
		self.prior_completeness = prior_completeness
		self.prob_ratio_secondary = prob_ratio_secondary
		self.device = device
		self.group = group
		self.capacity = capacity
		self.capacity_fixed = capacity  # the caller's choice stays (None: sized by the settling step, _build_plan)
		self.block_records_used = None
		self.tuning = tuning
		self.rank, self.world = world_info(group)
		self.plan = None
		self.setup()

	# -- hooks ---------------------------------------------------------------------------
	def _exchange_device(self):
		return self.device

	def _sync(self):
		import torch
		torch.cuda.synchronize(self.device)

	# -- one-time exchange ---------------------------------------------------------------
	def setup(self):
		import torch
		dist = _dist()
		t0 = time.perf_counter()
		dev = self._exchange_device()
		f64 = lambda x: torch.as_tensor(numpy.ascontiguousarray(numpy.asarray(x, dtype=float))).to(dev)
		# every rank needs ALL primaries: coordinates for the registration, errors for nothing but symmetry
		ra, counts = allgatherv(f64(self.primary['ra']), self.group, self.comm)
		dec, _ = allgatherv(f64(self.primary['dec']), self.group, self.comm)
		err, _ = allgatherv(f64(numpy.broadcast_to(numpy.asarray(self.primary['error'], dtype=float), numpy.shape(self.primary['ra']))), self.group, self.comm)
		self.gathered_bytes = 24 * int(ra.shape[0])
		self.primary_all = dict(name=self.primary['name'], ra=ra, dec=dec, error=err, area=self.primary['area'])
		self.bounds = numpy.concatenate([[0], numpy.cumsum(counts)]).astype(numpy.int64)
		self.primary_sizes = [int(c) for c in counts]
		self.primary_offset = int(self.bounds[self.rank])
		# global sizes and slice offsets of the secondaries
		self.sec_global, self.sec_offset = [], []
		for sl in self.secondary_slices:
			n = torch.tensor([len(sl['ra'])], dtype=torch.int64, device=dev)
			if self.world > 1:
				sizes = [torch.zeros_like(n) for _ in range(self.world)]
				dist.all_gather(sizes, n, group=self.group)
				sizes = [int(x.item()) for x in sizes]
			else:
				sizes = [int(n.item())]
			self.sec_global.append(sum(sizes))
			self.sec_offset.append(sum(sizes[:self.rank]))
		self._sync()
		self.setup_seconds = time.perf_counter() - t0
		# global indices travel as int32 (ExportRec, the idx columns): -1 is the 'absent' marker
		from nway_amd import _hip
		for n in [int(self.bounds[-1])] + self.sec_global:
			if n > _hip.CAPACITY_LIMIT:
				raise _hip.NwayHipError('a catalogue of %d rows exceeds the int32 index range of the match table' % n)
		self._decide()
		self._build_plan()

	def _decide(self):
		"""densities and cell scheme of the WHOLE catalogues (every rank must use the same)"""
		import torch
		import nway_amd
		from nway_amd import _hip
		log = nway_amd.NullOutputLogger()
		names = [self.primary['name']] + [s['name'] for s in self.secondary_slices]
		sizes = [int(self.bounds[-1])] + self.sec_global
		areas = [self.primary['area']] + [s['area'] for s in self.secondary_slices]
		self.dens, self.dens_plus = nway_amd._densities_from_sizes(names, sizes, areas, log)
		err = self.match_radius / 60. / 60
		local = [(numpy.asarray(self.primary['ra'], dtype=float), numpy.asarray(self.primary['dec'], dtype=float))]
		local += [(numpy.asarray(s['ra'], dtype=float), numpy.asarray(s['dec'], dtype=float)) for s in self.secondary_slices]
		scheme = nway_amd.choose_scheme([t for t in local if len(t[0]) > 0], err) if any(len(t[0]) for t in local) else _hip.SCHEME_FLAT
		if self.world > 1:
			s = torch.tensor([scheme], dtype=torch.int64, device=self._exchange_device())
			_dist().all_reduce(s, op=_dist().ReduceOp.MAX, group=self.group)  # the all-sky scheme wins
			scheme = int(s.item())
		self.scheme = scheme
		self.global_sizes = sizes

	def _build_plan(self):
		import ctypes
		import torch
		import nway_amd
		from nway_amd import _hip
		k = 1 + len(self.secondary_slices)
		err = self.match_radius / 60. / 60
		comp = nway_amd._completeness_vector(self.prior_completeness, k)
		self.params = _hip.make_params(k, self.scheme, self.match_radius, err, self.dens, self.dens_plus,
			nway_amd._prior_table(self.dens, self.dens_plus, comp), prob_ratio_secondary=self.prob_ratio_secondary, tuning=self.tuning)
		pa = self.primary_all
		self.cats = [_hip.DeviceCatalogue(pa['ra'], pa['dec'], pa['error'], self.device)]
		for s in self.secondary_slices:
			self.cats.append(_hip.DeviceCatalogue(s['ra'], s['dec'], numpy.asarray(s['error'], dtype=float), self.device))
		sizes = [c.n for c in self.cats]
		n_own = self.primary_sizes[self.rank]
		areas = [self.primary['area'] * 1.0] + [s['area'] * 1.0 for s in self.secondary_slices]
		_, cap_rows = nway_amd._estimate_capacities([max(n_own, 1)] + self.sec_global, areas, self.match_radius, self.scheme, True)
		self.bounds_dev = torch.as_tensor(self.bounds).to(self.device)
		capacity = self.capacity or max(1024, 4 * max(self.primary_sizes) // self.world + 1024)
		slot_retries = 0
		tightened = False
		for attempt in range(8):
			self.plan = _hip.MatchPlan(sizes, self.params, 65536, cap_rows, self.device, lean=True)
			if not self.plan.split_capable:
				# (decided from the plan alone, the same on every rank: nothing has been allocated or exchanged yet)
				self._drop_plan()
				raise _hip.NwayHipError('the secondary-split mode needs the sparse path with its one-launch fused tail: few chance '
					'neighbours per primary and at most 4 catalogues (3 with the unrelated-association correction); this plan is '
					'%d-way on path "%s". Shard by primary rows (ShardedMatch) or by declination zones (ZoneShardedMatch) instead'
					% (int(self.params.ncat), self.plan.description.get('tail', '?') if hasattr(self.plan, 'description') else '?'))
			nbytes = self.plan.split_buffer_bytes(self.world, capacity)
			self.export = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
			self.imported = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
			sp = _hip.Split()
			sp.world, sp.rank = self.world, self.rank
			sp.d_bounds = self.bounds_dev.data_ptr()
			sp.h_p_lo, sp.h_p_hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
			for c, off in enumerate(self.sec_offset):
				sp.slice_offset[c + 1] = off
			sp.capacity = capacity
			sp.d_export = self.export.data_ptr()
			sp.d_import = self.imported.data_ptr()
			self.split = sp
			self.capacity = capacity
			self.step()
			st = self.plan.read_status()
			flags = int(st[_hip.ST_FLAGS])
			# every rank must take the same decision: the capacities are part of the exchange's layout
			f = torch.tensor([flags, int(st[_hip.ST_SLOT_NEED])], dtype=torch.int64, device=self.device)
			if self.world > 1:
				allf = [torch.zeros_like(f) for _ in range(self.world)]
				_dist().all_gather(allf, f, group=self.group)
				flags_any, slot_need = 0, 0
				for x in allf:
					flags_any |= int(x[0].item())
					slot_need = max(slot_need, int(x[1].item()))
			else:
				flags_any, slot_need = flags, int(st[_hip.ST_SLOT_NEED])
			if flags_any == 0:
				# the exchange ships whole blocks: size them by what this settling step counted -- the fullest block of any
				# rank, twice that -- instead of the a-priori guess (5e5 x 1e8 over 8 ranks: 1 MB per peer for 0.2 MB of
				# records).  A later step on OTHER catalogues that outgrows it is flagged (PAIR_OVERFLOW: the receiver sees
				# count > capacity); ``resettle()`` then sizes the blocks again.
				used = self._fullest_block()
				tight = 2 * used + 64
				if not tightened and self.capacity_fixed is None and tight * 3 < capacity * 2:
					tightened = True
					capacity = tight
					self._drop_plan()
					continue
				self.status = st
				self.block_records_used = used
				return
			# the plan and what points into its buffers go before anything is raised or retried
			self._drop_plan()
			fatal = flags_any & (_hip.FLAG_LOOKBACK | _hip.FLAG_REG_OVERFLOW)
			if flags_any & _hip.FLAG_QUAD_DEEP:
				# (3-way: a primary with three candidates in a catalogue; the tail that walks them from now on, csrc/tail3q.inc)
				self.params.disable = int(self.params.disable) | _hip.DISABLE_QUAD3
			if flags_any & _hip.FLAG_SLOT_OVERFLOW and not fatal and 0 < slot_need <= _hip.LINK_SLOTS_MAX_FUSED and slot_retries < 2:
				# a clustered primary: once or twice more with the slots the run counted (as run_plan does)
				slot_retries += 1
				self.params.link_slots = min(_hip.LINK_SLOTS_MAX_FUSED, slot_need + (slot_need >> 3) + 1)
			elif flags_any & _hip.FLAG_SLOT_OVERFLOW or fatal:
				raise _hip.NwayHipError('the secondary-split mode does not fit this input (status flags %d): shard by primary rows' % flags_any)
			if flags_any & _hip.FLAG_PAIR_OVERFLOW:
				capacity *= 4
			if flags_any & _hip.FLAG_ROW_OVERFLOW:
				cap_rows = min(cap_rows * 2, (1 << 31) - 4096)
		raise _hip.NwayHipError('secondary-split mode: capacities could not be settled')

	def _fullest_block(self):
		"""records in the fullest (source, catalogue) block any rank received in the last step (header record of every block:
		sparse.inc, ExportRec), the same number on every rank"""
		import torch
		k1 = len(self.secondary_slices)
		words = self.imported.view(torch.int32)[:self.world * k1 * (self.capacity + 1) * 8]
		heads = words.view(self.world * k1, (self.capacity + 1) * 8)[:, 0]
		m = heads.max().to(torch.int64).reshape(1)
		if self.world > 1:
			_dist().all_reduce(m, op=_dist().ReduceOp.MAX, group=self.group)
		return int(m.item())

	def _drop_plan(self):
		"""close the plan and forget everything that points into its buffers"""
		if self.plan is not None:
			self.plan.close()
		self.plan = None
		self.split = None
		self.export = None
		self.imported = None

	# -- per step ------------------------------------------------------------------------
	def _exchange(self):
		dist = _dist()
		if self.comm is not None:
			# the library's own RCCL group of send / recv pairs on the pipeline's stream (nwayhip_comm_exchange): no host
			# round trip between the two halves
			self.comm.exchange(self.export, self.imported)
		elif self.world == 1:
			self.imported.copy_(self.export)
		elif dist.get_backend(self.group) == 'nccl':
			dist.all_to_all_single(self.imported, self.export, group=self.group)
		else:
			# gloo (functional tests, several ranks sharing one GPU): through host memory
			src = self.export.cpu()
			dst = src.clone()
			dist.all_to_all_single(dst, src, group=self.group)
			self.imported.copy_(dst)

	def step(self):
		"""one pass: register + sweep of the own slices, ONE all-to-all, import + fused tail"""
		self.plan.split_front(self.cats, self.split)
		self._exchange()
		self.plan.split_back(self.cats, self.split)

	def resettle(self):
		"""COLLECTIVE: size the export blocks, the slots and the row capacity again (the settling loop of the set-up).  For a
		caller whose ``step()`` came back with FLAG_PAIR_OVERFLOW / FLAG_ROW_OVERFLOW in ``read_status()`` after the slices
		changed under the plan; every rank must call it."""
		self._drop_plan()
		self._build_plan()

	def read_status(self):
		if self.plan is None:
			raise RuntimeError('SecondarySplitMatch: no plan (set-up failed)')
		return self.plan.read_status()

	def local_rows(self):
		from nway_amd import _hip
		return int(self.read_status()[_hip.ST_ROWS])

	def total_rows(self):
		import torch
		n = self.local_rows()
		if self.world == 1:
			return n
		t = torch.tensor([n], dtype=torch.int64, device=self._exchange_device())
		_dist().all_reduce(t, group=self.group)
		return int(t.item())

	def pass_bytes(self, rows):
		"""algorithmic bytes of this rank's pass: all the primaries it registers, its slices
		of the secondaries (ra, dec; the error where it is a column) + the output columns of its rows"""
		k = 1 + len(self.secondary_slices)
		b = int(self.bounds[-1]) * 16.0 + self.primary_sizes[self.rank] * 8.0
		for s in self.secondary_slices:
			b += len(s['ra']) * (16.0 + (8.0 if numpy.ndim(s['error']) > 0 else 0.0))
		per_row = 4 * k + 8 * (k * (k - 1) // 2) + 8 + 1 + 8 * 5 + 1
		return b + per_row * rows

	def local_table(self):
		"""this rank's block of the global table as host columns (global indices throughout)"""
		from nway_amd import _hip
		st = self.plan.read_status()
		m = int(st[_hip.ST_ROWS])
		names = [self.primary['name']] + [s['name'] for s in self.secondary_slices]
		t = {}
		for c, nme in enumerate(names):
			t[nme] = _hip.to_host(self.plan.cols['idx'][c][:m]).astype(numpy.int64)
		for p, (i, j) in enumerate(_hip.pair_columns(len(names))):
			t['Separation_%s_%s' % (names[i], names[j])] = _hip.to_host(self.plan.cols['sep'][p][:m])
		for src, dst in (('sep_max', 'Separation_max'), ('log_bf', 'dist_bayesfactor_uncorrected'), ('log_bf_corrected', 'dist_bayesfactor'),
				('dist_post', 'dist_post'), ('p_single', 'p_single'), ('p_any', 'prob_has_match'), ('p_i', 'prob_this_match')):
			t[dst] = _hip.to_host(self.plan.cols[src][:m])
		t['ncat'] = _hip.to_host(self.plan.cols['ncat'][:m]).astype(numpy.int64)
		t['match_flag'] = _hip.to_host(self.plan.cols['match_flag'][:m]).astype(numpy.int64)
		return t

	def gather_table(self, dst=0):
		"""global table on rank ``dst`` (rank-order concatenation); None elsewhere"""
		dist = _dist()
		local = self.local_table()
		if self.world == 1:
			return local
		gathered = [None] * self.world if self.rank == dst else None
		dist.gather_object(local, gathered, dst=dst, group=self.group)
		if self.rank != dst:
			return None
		out = {}
		for key in gathered[0]:
			if key.startswith('_'):
				continue
			out[key] = numpy.concatenate([numpy.asarray(g[key]) for g in gathered])
		return out


def exchange_rows(rows, send_counts, group=None):
	"""all-to-all-v of the ROWS of a 2-D float64 tensor: ``rows`` sorted by destination rank, ``send_counts[r]`` of them for
	rank r.  Returns what arrived, in source-rank order (the order of the rows of one source is kept).  RCCL: two
	``all_to_all_single`` (the counts, then the rows with split sizes) on device tensors; gloo: the same calls on host
	tensors (device tensors -- ranks sharing one GPU in the functional tests -- go through the host)."""
	import torch
	dist = _dist()
	if len(send_counts) == 1:  # (one destination: this process alone, whatever process group is up)
		return rows
	rank, world = world_info(group)
	if world == 1:
		return rows
	on_host = dist.get_backend(group) != 'nccl'
	dev = rows.device
	if on_host:
		rows = rows.cpu()
	sc = torch.as_tensor([int(c) for c in send_counts], dtype=torch.int64, device=rows.device)
	rc = torch.zeros_like(sc)
	dist.all_to_all_single(rc, sc, group=group)
	recv_counts = [int(c) for c in rc.tolist()]
	out = torch.empty((sum(recv_counts), rows.shape[1]), dtype=rows.dtype, device=rows.device)
	dist.all_to_all_single(out, rows.contiguous(), recv_counts, [int(c) for c in send_counts], group=group)
	return out.to(dev) if on_host else out


class ZoneShardedMatch(MagnitudePriors):
	"""ONE job over several GPUs with BOTH sides sharded by declination zones (what BASELINE configs[4] calls pre-bucketing,
	done across the GPUs): rank z owns the primaries whose declination lies in zone z and holds the secondaries within the
	match radius of that zone -- a great-circle separation is at least the difference of the declinations, so every
	counterpart of an owned primary is there.  The catalogues are redistributed ONCE at set-up (an all-to-all-v of rows:
	every source travels to one rank, the secondaries inside the seams to two -- an eighth of what the all-gatherv of
	``ShardedMatch`` moves to every rank); no collective is on the per-step path, and a step streams 1/N of the
	secondaries instead of all of them (``ShardedMatch``) without registering all primaries on every rank or routing
	candidates between the ranks every step (``SecondarySplitMatch``).  Zone edges: quantiles of the declinations of the
	largest secondary catalogue (a histogram summed over the ranks), so every rank streams the same number of sources.

	Every row of the table belongs to one primary, its values depend on that primary, its candidates and the densities of
	the WHOLE catalogues only: a rank's rows are bit-identical to the same rows of the unsharded run.  The rows of a rank are
	ordered by global primary index (the redistribution keeps the order of the rows of a source rank, and the source ranks
	hold ascending ranges), with global indices in the index columns; the global table is the concatenation of the ranks'
	tables sorted (stably) by primary -- zones interleave in index space, so unlike the other two modes the rank order
	alone is not the global order (``gather_table`` sorts).

	primary: this rank's shard of the primary catalogue (contiguous global rows, rank order); secondaries: list of this
	rank's SLICES (contiguous global rows) of the secondary catalogues; ``error`` may be a scalar.
	Hooks as in ``ShardedMatch`` (``_exchange_device``, ``_sync``, ``_build_plan``, ``step``, ``local_rows``,
	``_local_columns``): the HIP pipeline here, the oracle in the CPU tests.
	"""

	ZONE_BINS = 1 << 16
	CUT_WHERE_RESIDENT = True   # (tests: False = the host-side cut also where this process is alone)

	def __init__(self, primary, secondaries, match_radius, prior_completeness, device, group=None,
			prob_ratio_secondary=0.5, tuning=None, comm=None, zones_per_rank=1, streams=1, local_only=False, one_launch=True, registration='auto'):
		"""local_only: this process alone, whatever process group is up (the catalogues handed in are the WHOLE catalogues; no collective
		is issued) -- ``bench.py`` measures the one-GPU reference of a job on rank 0 that way while the other ranks wait.
		zones_per_rank: every rank holds this many declination zones and runs them one after the other in a step (round 5).
		A zone is an ordinary match of its own: several small zones keep the cell table of each within the LDS of a sweep
		workgroup where the one big zone would need the large-table sweep (5e5 x 1e8 on ONE GPU: 729 us as one zone) -- and with
		``streams`` > 1 the zones of a step are enqueued round robin on that many HIP streams, so that the latency-bound ends of
		one zone's pass (registration, routing chain, tail) run beside the stream of another's.  The table does not depend on
		either number.
		one_launch (round 6; with ``streams`` == 1): the zones of a step go out as ONE launch set -- one registration, one sweep,
		one tail launch for all of them (``_hip.ZoneBatch``, include/nwayhip.h: nwayhip_zones_*) -- instead of three launches per
		zone: the fixed latencies of a pass are paid once per step, not once per zone.  Where the zones' plans do not qualify
		(2-way sparse tail, table in LDS) the library enqueues them one after the other; ``batched`` tells."""
		if isinstance(secondaries, dict):
			secondaries = [secondaries]
		if comm not in (None, 'torch'):
			# the set-up exchange is an all-to-all-v of rows through torch.distributed and nothing travels per step: there is
			# no carrier to choose (the other two modes take comm='rccl')
			raise ValueError("ZoneShardedMatch has no per-step exchange: comm must be None (got %r)" % (comm,))
		self.comm = None
		self.primary = primary
		self.secondary_slices = secondaries
		self.match_radius = float(match_radius)
		self.prior_completeness = prior_completeness
		self.prob_ratio_secondary = prob_ratio_secondary
		self.device = device
		self.group = group
		self.tuning = tuning
		self.rank, self.world = (0, 1) if local_only else world_info(group)
		self.zones_per_rank = max(1, int(zones_per_rank))
		self.nstreams = max(1, int(streams))
		self.one_launch = bool(one_launch)
		self.registration = registration   # of a launch set: 'auto' | 'atomics' | 'owner' (_hip.ZoneBatch)
		self._batch = None
		self.batched = False
		self.owner_computes = False
		self.plan = None
		self.setup_seconds = None
		self.setup()

	# -- hooks ---------------------------------------------------------------------------
	def _exchange_device(self):
		return self.device

	def _sync(self):
		import torch
		torch.cuda.synchronize(self.device)

	# -- one-time exchange ---------------------------------------------------------------
	def _global_sizes(self, n):
		"""(sizes on every rank, in rank order) of a local row count"""
		import torch
		if self.world == 1:
			return [int(n)]
		t = torch.tensor([int(n)], dtype=torch.int64, device=self._exchange_device())
		sizes = [torch.zeros_like(t) for _ in range(self.world)]
		_dist().all_gather(sizes, t, group=self.group)
		return [int(s.item()) for s in sizes]

	def _zone_edges(self, resident=None):
		"""interior edges (world x zones_per_rank - 1 of them, ascending) of the declination zones: equal shares of the largest secondary catalogue.
		resident: this process alone -- the catalogues' columns as ``setup`` has uploaded them; the histogram is then taken where they lie
		(the same arithmetic, hence the same edges: a hundred million declinations are seconds of numpy on the host)"""
		import torch
		dist = _dist()
		dev = self._exchange_device()
		big = int(numpy.argmax(self.sec_global)) if self.sec_global else 0
		nz = self.world * self.zones_per_rank
		if resident is not None and self.secondary_slices:
			d = resident[1 + big][1]
			d = d[torch.isfinite(d)]
			lo, hi = (float(d.min().item()), float(d.max().item())) if d.numel() else (numpy.inf, -numpy.inf)
			if not (hi > lo):
				return numpy.full(nz - 1, lo if numpy.isfinite(lo) else 0.0)
			width = (hi - lo) / self.ZONE_BINS
			h = torch.bincount(torch.clamp(((d - lo) / width).to(torch.int64), max=self.ZONE_BINS - 1), minlength=self.ZONE_BINS)
			return self._edges_from_histogram(h.cpu().numpy(), lo, width, nz)
		dec = numpy.asarray(self.secondary_slices[big]['dec'], dtype=float) if self.secondary_slices else numpy.zeros(0)
		dec = dec[numpy.isfinite(dec)]
		lim = torch.tensor([dec.min() if len(dec) else numpy.inf, -(dec.max() if len(dec) else -numpy.inf)], dtype=torch.float64, device=dev)
		if self.world > 1:
			dist.all_reduce(lim, op=dist.ReduceOp.MIN, group=self.group)
		lo, hi = float(lim[0].item()), -float(lim[1].item())
		if not (hi > lo):
			return numpy.full(nz - 1, lo if numpy.isfinite(lo) else 0.0)
		width = (hi - lo) / self.ZONE_BINS
		hist = numpy.bincount(numpy.minimum(((dec - lo) / width).astype(numpy.int64), self.ZONE_BINS - 1), minlength=self.ZONE_BINS).astype(numpy.int64)
		h = torch.as_tensor(hist).to(dev)
		if self.world > 1:
			dist.all_reduce(h, group=self.group)
		return self._edges_from_histogram(h.cpu().numpy(), lo, width, nz)

	@staticmethod
	def _edges_from_histogram(hist, lo, width, nz):
		cum = numpy.cumsum(hist)
		total = int(cum[-1])
		edges = []
		for z in range(1, nz):
			b = int(numpy.searchsorted(cum, (total * z) // nz, side='left'))
			edges.append(lo + (b + 1) * width)
		return numpy.maximum.accumulate(numpy.asarray(edges, dtype=float))

	def setup(self):
		import torch
		t0 = time.perf_counter()
		dev = self._exchange_device()
		world = self.world
		self.primary_sizes = self._global_sizes(len(self.primary['ra']))
		self.primary_offset = int(sum(self.primary_sizes[:self.rank]))
		self.sec_global, self.sec_offset = [], []
		for sl in self.secondary_slices:
			sizes = self._global_sizes(len(sl['ra']))
			self.sec_global.append(sum(sizes))
			self.sec_offset.append(sum(sizes[:self.rank]))
		from nway_amd import _hip
		for n in [sum(self.primary_sizes)] + self.sec_global:
			if n > _hip.CAPACITY_LIMIT:
				raise _hip.NwayHipError('a catalogue of %d rows exceeds the int32 index range of the match table' % n)
		# This process alone (one GPU, or ``local_only``): nothing travels between ranks, so the catalogues go up ONCE, as they are, and
		# are cut into zones where they lie -- the host-side cut below (masks, gathers and a packed copy of a hundred million rows, then a
		# pageable upload) was 6 of the 6.7 s of set-up of BASELINE configs[4] on one GPU.  Same rows in the same order either way.
		resident = None
		if world == 1 and self.CUT_WHERE_RESIDENT:
			resident = []
			for t in [self.primary] + list(self.secondary_slices):
				cols = [t['ra'], t['dec']] + ([] if numpy.ndim(t['error']) == 0 else [t['error']])
				resident.append(_hip.upload_columns(cols, dev) if torch.device(dev).type == 'cuda' else [_hip.to_device(c, dev) for c in cols])  # (the CPU tests' engines)
		self.edges = self._zone_edges(resident)
		margin = self.match_radius / 3600. * (1 + 1e-9) + 1e-12
		self.moved_bytes = 0

		zpr = self.zones_per_rank
		nz = world * zpr

		def cut_resident(table, cols, offset, seams):
			"""``redistribute`` for this process alone: the zones of ``table`` out of its resident columns ``cols``"""
			scalar_error = numpy.ndim(table['error']) == 0
			ra, dec = cols[0], cols[1]
			edges = torch.as_tensor(self.edges, dtype=torch.float64, device=dec.device)
			m = margin if seams else 0.0
			# (numpy.searchsorted(edges, x, side='right') of the host-side cut)
			z_lo = torch.bucketize(dec - m, edges, right=True)
			z_hi = torch.bucketize(dec + m, edges, right=True)
			bad = ~torch.isfinite(dec)
			z_lo[bad] = nz - 1
			z_hi[bad] = nz - 1
			out = []
			for zl in range(zpr):
				rows = torch.nonzero((z_lo <= zl) & (zl <= z_hi)).reshape(-1)  # (ascending: local order = global order)
				t = dict(name=table['name'], ra=ra.index_select(0, rows), dec=dec.index_select(0, rows), area=table['area'],
					error=(float(table['error']) if scalar_error else cols[2].index_select(0, rows)), mags=[], maghists=[], magnames=[])
				self.moved_bytes += int(rows.numel()) * 8 * (len(cols) + 1 + (1 if zpr > 1 else 0))  # (what the packed rows of the exchange would hold)
				out.append((t, (rows + offset).cpu().numpy().astype(numpy.int64)))
			return out

		def redistribute(table, offset, seams):
			"""rows of ``table`` -> the ranks of their zones; returns per LOCAL zone (host-side bookkeeping aside, the columns stay where
			they arrived) a table dict and the global indices of its rows"""
			ra = numpy.asarray(table['ra'], dtype=float)
			dec = numpy.asarray(table['dec'], dtype=float)
			scalar_error = numpy.ndim(table['error']) == 0
			z_lo = numpy.searchsorted(self.edges, dec - (margin if seams else 0.0), side='right')
			z_hi = numpy.searchsorted(self.edges, dec + (margin if seams else 0.0), side='right')
			bad = ~numpy.isfinite(dec)
			z_lo[bad] = z_hi[bad] = nz - 1  # (a source without a declination matches nothing; a primary still has its row)
			picks, counts, zone_of = [], [0] * world, []
			for z in range(nz):
				rows = numpy.flatnonzero((z_lo <= z) & (z <= z_hi))
				picks.append(rows)
				zone_of.append(numpy.full(len(rows), float(z)))
				counts[z // zpr] += len(rows)
			rows = numpy.concatenate(picks) if picks else numpy.zeros(0, dtype=numpy.int64)
			cols = [ra[rows], dec[rows]] + ([] if scalar_error else [numpy.asarray(table['error'], dtype=float)[rows]]) + [(rows + offset).astype(float)]
			if zpr > 1:
				cols.append(numpy.concatenate(zone_of) if zone_of else numpy.zeros(0))
			packed = torch.as_tensor(numpy.ascontiguousarray(numpy.stack(cols, axis=1))).to(dev)
			self.moved_bytes += int(packed.numel()) * 8
			got = exchange_rows(packed, counts, self.group)
			out = []
			for zl in range(zpr):
				if zpr > 1:
					# (arrival order = source rank, then zone, then the source's row order: inside ONE zone the rows are still ascending in
					# the global index, which is what makes local order = global order)
					sel = torch.nonzero(got[:, -1] == float(self.rank * zpr + zl)).reshape(-1)
					part = got.index_select(0, sel)
				else:
					part = got
				g_col = -2 if zpr > 1 else -1
				def column(i):
					# (a column of a one-row table is already "contiguous" where it lies inside the packed row: 8-byte aligned only --
					# found by the local-zones soak, where a zone may hold one source of a catalogue; the library wants 16)
					c = part[:, i].contiguous()
					return c.clone() if c.data_ptr() % 16 else c
				t = dict(name=table['name'], ra=column(0), dec=column(1), area=table['area'],
					error=(float(table['error']) if scalar_error else column(2)), mags=[], maghists=[], magnames=[])
				out.append((t, part[:, g_col].cpu().numpy().astype(numpy.int64)))
			return out
		if resident is not None:
			prim = cut_resident(self.primary, resident[0], self.primary_offset, False)
			secs = [cut_resident(sl, cols, off, True) for sl, cols, off in zip(self.secondary_slices, resident[1:], self.sec_offset)]
			del resident
		else:
			prim = redistribute(self.primary, self.primary_offset, False)
			secs = [redistribute(sl, off, True) for sl, off in zip(self.secondary_slices, self.sec_offset)]
		self.zones = []
		for zl in range(zpr):
			self.zones.append(dict(primary=prim[zl][0], primary_gidx=prim[zl][1], secondaries=[sc[zl][0] for sc in secs], sec_gidx=[sc[zl][1] for sc in secs],
				plan=None, cats=None, empty=True, status=None))
		# (one zone per rank: the names of round 4)
		self.zone_primary, self.primary_gidx = self.zones[0]['primary'], self.zones[0]['primary_gidx']
		self.zone_secondaries, self.sec_gidx = self.zones[0]['secondaries'], self.zones[0]['sec_gidx']
		self._sync()
		self.setup_seconds = time.perf_counter() - t0
		self._decide()
		self._build_plan()

	def _decide(self):
		"""densities and cell scheme of the WHOLE catalogues (every rank must use the same)"""
		import torch
		import nway_amd
		from nway_amd import _hip
		log = nway_amd.NullOutputLogger()
		names = [self.primary['name']] + [s['name'] for s in self.secondary_slices]
		self.global_sizes = [int(sum(self.primary_sizes))] + self.sec_global
		areas = [self.primary['area']] + [s['area'] for s in self.secondary_slices]
		self.dens, self.dens_plus = nway_amd._densities_from_sizes(names, self.global_sizes, areas, log)
		err = self.match_radius / 60. / 60
		local = [(numpy.asarray(self.primary['ra'], dtype=float), numpy.asarray(self.primary['dec'], dtype=float))]
		local += [(numpy.asarray(s['ra'], dtype=float), numpy.asarray(s['dec'], dtype=float)) for s in self.secondary_slices]
		scheme = nway_amd.choose_scheme([t for t in local if len(t[0]) > 0], err) if any(len(t[0]) for t in local) else _hip.SCHEME_FLAT
		if self.world > 1:
			s = torch.tensor([scheme], dtype=torch.int64, device=self._exchange_device())
			_dist().all_reduce(s, op=_dist().ReduceOp.MAX, group=self.group)  # the all-sky scheme wins
			scheme = int(s.item())
		self.scheme = scheme

	def _build_plan(self):
		"""per local zone: device catalogues and a settled plan (hook ``_build_zone``)"""
		import nway_amd
		from nway_amd import _hip
		k = 1 + len(self.secondary_slices)
		err = self.match_radius / 60. / 60
		comp = nway_amd._completeness_vector(self.prior_completeness, k)
		self.params = _hip.make_params(k, self.scheme, self.match_radius, err, self.dens, self.dens_plus,
			nway_amd._prior_table(self.dens, self.dens_plus, comp), prob_ratio_secondary=self.prob_ratio_secondary, tuning=self.tuning)
		for z in self.zones:
			self._build_zone(z)
		self._streams = None
		if getattr(self, '_batch', None) is not None:
			for b, _ in self._batch:
				b.close()
		self._batch = None
		z0 = self.zones[0]
		self.plan, self.cats, self.empty, self.status = z0['plan'], z0['cats'], all(z['empty'] for z in self.zones), z0['status']

	def _build_zone(self, z):
		import nway_amd
		from nway_amd import _hip
		tables = [z['primary']] + z['secondaries']
		z['empty'] = len(z['primary']['ra']) == 0  # (a zone without primaries has no rows)
		z['cats'] = [_hip.DeviceCatalogue(t['ra'], t['dec'], t['error'], self.device) for t in tables]  # (tensors as they arrived; a scalar error stays one)
		if z['empty']:
			z['plan'], z['status'] = None, numpy.zeros(_hip.STATUS_WORDS, dtype=numpy.int64)
			return
		sizes = [c.n for c in z['cats']]
		# (capacities from the densities of the whole job: the zone's share of the sky is not known to the estimate)
		areas = [t['area'] * max(n, 1) / max(g, 1) for t, n, g in zip(tables, sizes, self.global_sizes)]
		cap_pairs, cap_rows = nway_amd._estimate_capacities(sizes, areas, self.match_radius, self.scheme, True)
		z['plan'], z['status'] = _hip.run_plan(sizes, type(self.params).from_buffer_copy(self.params), z['cats'], cap_pairs, cap_rows, self.device, lean=True)

	# -- per step ------------------------------------------------------------------------
	def step(self, cats=None):
		"""one pass of the hot path over this rank's zone(s) (no collective).  Several zones: one after the other, or -- ``streams`` > 1 --
		round robin on that many streams, which start behind the work already on the current stream and which the current stream
		waits for at the end (whatever follows on it sees the finished tables)"""
		live = [z for z in self.zones if not z['empty']]
		if cats is not None and len(self.zones) > 1:
			raise ValueError('ZoneShardedMatch.step(cats=...) stands for the catalogues of ONE zone: this rank holds %d' % len(self.zones))
		if len(live) > 1 and self.one_launch:
			from nway_amd import _hip
			import torch
			# one launch set -- or, ``streams`` > 1, that many launch sets of consecutive zones on streams of their own: the
			# registration of one set (bound by memory-side atomics, a wave per SIMD) then runs beside the sweep of another
			ngroups = min(self.nstreams, len(live) // 2) if self.nstreams > 1 else 1
			if self._batch is None:
				bounds = [len(live) * g // max(ngroups, 1) for g in range(max(ngroups, 1) + 1)]
				self._batch = [(_hip.ZoneBatch([z['plan'] for z in live[lo:hi]], registration=self.registration), [z['cats'] for z in live[lo:hi]])
					for lo, hi in zip(bounds[:-1], bounds[1:])]
			if len(self._batch) == 1:
				self._batch[0][0].enqueue(self._batch[0][1])
			else:
				if self._streams is None:
					self._streams = side_streams(self.device, len(self._batch))
				cur = torch.cuda.current_stream(self.device)
				for st in self._streams:
					st.wait_stream(cur)
				for (batch, cats_of), st in zip(self._batch, self._streams):
					batch.enqueue(cats_of, stream=ctypes_stream(st))
				for st in self._streams:
					cur.wait_stream(st)
			self.batched = all(b.batched for b, _ in self._batch)
			self.owner_computes = all(b.owner_computes for b, _ in self._batch)
			return
		if len(live) <= 1 or self.nstreams == 1:
			for z in live:
				self._step_zone(z, cats, None)
			return
		import torch
		if self._streams is None:
			self._streams = side_streams(self.device, min(self.nstreams, len(live)))
		cur = torch.cuda.current_stream(self.device)
		for st in self._streams:
			st.wait_stream(cur)
		for i, z in enumerate(live):
			self._step_zone(z, None, self._streams[i % len(self._streams)])
		for st in self._streams:
			cur.wait_stream(st)

	def _step_zone(self, z, cats, stream):
		if stream is None:
			z['plan'].enqueue(z['cats'] if cats is None else cats)
		else:
			z['plan'].enqueue(z['cats'], stream=ctypes_stream(stream))

	def _zone_status(self, z):
		return z['status'] if z['plan'] is None else z['plan'].read_status()

	def read_status(self):
		"""the status words of the rank's pass; several zones: rows and counters summed, flags or'ed"""
		from nway_amd import _hip
		sts = [numpy.asarray(self._zone_status(z)) for z in self.zones]
		if len(sts) == 1:
			return sts[0]
		out = numpy.sum(sts, axis=0)
		flags = 0
		for st in sts:
			flags |= int(st[_hip.ST_FLAGS])
		out[_hip.ST_FLAGS] = flags
		return out

	def local_rows(self):
		from nway_amd import _hip
		return int(self.read_status()[_hip.ST_ROWS])

	def total_rows(self):
		import torch
		n = self.local_rows()
		if self.world == 1:
			return n
		t = torch.tensor([n], dtype=torch.int64, device=self._exchange_device())
		_dist().all_reduce(t, group=self.group)
		return int(t.item())

	def close(self):
		if getattr(self, '_batch', None) is not None:
			for b, _ in self._batch:
				b.close()
			self._batch = None
		for z in self.zones:
			if z.get('plan') is not None:
				z['plan'].close()
				z['plan'] = None
		self.plan = None

	def pass_bytes(self, rows):
		"""algorithmic bytes of this rank's pass: the primaries and secondaries of its zone(s) read once + its rows"""
		k = 1 + len(self.secondary_slices)
		b = 0.0
		for z in self.zones:
			for t in [z['primary']] + z['secondaries']:
				b += len(t['ra']) * (16.0 + (8.0 if hasattr(t['error'], 'shape') and len(t['error'].shape) > 0 else 0.0))
		return b + (4 * k + 8 * (k * (k - 1) // 2) + 8 + 1 + 8 * 5 + 1) * rows

	def _zone_columns(self, z):
		"""the rows of one zone as host columns, indices LOCAL to the zone"""
		from nway_amd import _hip
		names = [self.primary['name']] + [s['name'] for s in self.secondary_slices]
		if z['plan'] is None:
			t = dict((nme, numpy.zeros(0, dtype=numpy.int64)) for nme in names + ['ncat', 'match_flag'])
			for i, j in _hip.pair_columns(len(names)):
				t['Separation_%s_%s' % (names[i], names[j])] = numpy.zeros(0)
			for dst in ('Separation_max', 'dist_bayesfactor_uncorrected', 'dist_bayesfactor', 'dist_post', 'p_single', 'prob_has_match', 'prob_this_match'):
				t[dst] = numpy.zeros(0)
			return t
		plan = z['plan']
		m = int(plan.read_status()[_hip.ST_ROWS])
		t = {}
		for c, nme in enumerate(names):
			t[nme] = _hip.to_host(plan.cols['idx'][c][:m]).astype(numpy.int64)
		for p, (i, j) in enumerate(_hip.pair_columns(len(names))):
			t['Separation_%s_%s' % (names[i], names[j])] = _hip.to_host(plan.cols['sep'][p][:m])
		for src, dst in (('sep_max', 'Separation_max'), ('log_bf', 'dist_bayesfactor_uncorrected'), ('log_bf_corrected', 'dist_bayesfactor'),
				('dist_post', 'dist_post'), ('p_single', 'p_single'), ('p_any', 'prob_has_match'), ('p_i', 'prob_this_match')):
			t[dst] = _hip.to_host(plan.cols[src][:m])
		t['ncat'] = _hip.to_host(plan.cols['ncat'][:m]).astype(numpy.int64)
		t['match_flag'] = _hip.to_host(plan.cols['match_flag'][:m]).astype(numpy.int64)
		return t

	def _local_columns(self):
		"""(one zone per rank, the hook of round 4) the rows of this rank's zone, LOCAL indices"""
		return self._zone_columns(self.zones[0])

	def local_table(self):
		"""this rank's rows as host columns, GLOBAL indices throughout, ascending in the primary (several zones: their tables
		concatenated and sorted stably by primary -- a primary lives in one zone)"""
		names = [self.primary['name']] + [s['name'] for s in self.secondary_slices]
		parts = []
		for zi, z in enumerate(self.zones):
			t = dict(self._local_columns() if len(self.zones) == 1 else self._zone_columns(z))
			for nme, g in zip(names, [z['primary_gidx']] + z['sec_gidx']):
				col = numpy.asarray(t[nme], dtype=numpy.int64)
				t[nme] = numpy.where(col >= 0, g[numpy.maximum(col, 0)], -1) if len(g) else col
			parts.append(t)
		if len(parts) == 1:
			return parts[0]
		fullest = max(parts, key=lambda t: len(t[names[0]]))
		live = [t for t in parts if len(t[names[0]]) > 0] or [fullest]
		out = dict((key, numpy.concatenate([numpy.asarray(t[key]) for t in live])) for key in fullest if not key.startswith('_'))
		order = numpy.argsort(out[names[0]], kind='stable')
		return dict((key, col[order]) for key, col in out.items())

	def gather_table(self, dst=0):
		"""global table on rank ``dst``: the ranks' tables concatenated and sorted (stably) by primary; None elsewhere"""
		dist = _dist()
		local = self.local_table()
		gathered = [local]
		if self.world > 1:
			gathered = [None] * self.world if self.rank == dst else None
			dist.gather_object(local, gathered, dst=dst, group=self.group)
			if self.rank != dst:
				return None
		out = {}
		fullest = max(gathered, key=lambda g: len(g[self.primary['name']]))  # (a zone without primaries may know fewer columns)
		for key in fullest:
			if not key.startswith('_'):
				out[key] = numpy.concatenate([numpy.asarray(g[key]) for g in gathered if key in g and len(g[self.primary['name']]) > 0] or [numpy.asarray(fullest[key])[:0]])
		order = numpy.argsort(out[self.primary['name']], kind='stable')
		return dict((key, col[order]) for key, col in out.items())
