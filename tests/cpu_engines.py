"""CPU stand-ins for the device halves of nway_amd.distributed, for the gloo tests (no GPU here).

The product classes keep the exchange logic -- sharding, all-gatherv, global indices, the
candidate all-to-all, rank-order concatenation -- behind a handful of hooks that are the HIP
pipeline in the package.  These subclasses fill the hooks with the oracle (test infrastructure):
what is under test is everything in nway_amd.distributed EXCEPT the hooks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from goldenutil import ROOT

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from nway_amd import distributed  # noqa: E402


class NumpyMagnitudeHooks(object):
	"""the two per-row hooks of distributed.MagnitudePriors in numpy (the HIP engines run nwayhip_bias_lookup / nwayhip_group_stats)"""

	def _bias_lookup(self, idx, mag_all, func, total):
		idx = np.asarray(idx, dtype=np.int64)
		m = np.where(idx >= 0, mag_all[np.maximum(idx, 0)] if len(mag_all) else np.nan, np.nan)
		with np.errstate(divide='ignore'):
			w = np.log10(func(m))
		w[np.isnan(w)] = 0
		total += w
		return 10**w

	def _final_probabilities(self, primary_index, ncat, total, prior, ratio):
		import nway_oracle as orc
		if len(total) == 0:
			return np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0, dtype=np.int64)
		p_single = orc.posterior(prior, total)
		p_any, p_i, flag = orc.group_statistics(np.asarray(primary_index), orc.unnormalised_log_posterior(prior, total, np.asarray(ncat)), ratio)
		return p_single, p_any, p_i, flag


class OracleShardedMatch(NumpyMagnitudeHooks, distributed.ShardedMatch):
	"""primary rows sharded, secondaries all-gathered; the per-rank match is the numpy oracle"""

	def _exchange_device(self):
		return torch.device('cpu')

	def _sync(self):
		pass

	def _build_plan(self):
		self.empty = len(self.primary['ra']) == 0
		self.table = None

	def step(self):
		import nway_oracle as orc
		tables = self._tables()
		tables = [dict(t, ra=np.asarray(t['ra']), dec=np.asarray(t['dec']),
			error=(t['error'] if np.ndim(t['error']) == 0 else np.asarray(t['error']))) for t in tables]
		self.table = orc.nway_match(tables, self.match_radius, self.prior_completeness, prob_ratio_secondary=self.prob_ratio_secondary)
		return self.table

	def local_rows(self):
		return len(self.table['ncat'])

	def local_table(self):
		t = dict(self.table)
		pname = self.primary['name']
		t[pname] = np.asarray(t[pname]) + self.primary_offset
		return t


def split_front(primary_all, slices, radius, scheme):
	"""what the device front half exports: per catalogue the candidates (primary, secondary of the slice)"""
	import nway_oracle as orc
	out = []
	for sl in slices:
		tup = orc.enumerate_tuples([(primary_all['ra'], primary_all['dec']), (sl['ra'], sl['dec'])], radius / 3600., scheme, radius)
		tup = tup[tup[:, 1] >= 0]
		out.append((tup[:, 0], tup[:, 1]))
	return out


def split_back(own, p_lo, received, names, areas, radius, completeness, densities, scheme, ratio):
	"""what the device back half does with the records it received: the match of the own primaries
	against exactly those secondaries, with the densities and the scheme of the whole catalogues"""
	import nway_oracle as orc
	tables = [dict(name=own['name'], ra=own['ra'], dec=own['dec'], error=np.broadcast_to(np.asarray(own['error'], dtype=float), np.shape(own['ra'])), area=own['area'])]
	gidx = []
	for c, rec in enumerate(received):
		g, first = np.unique(rec[:, 1].astype(np.int64), return_index=True)  # ascending global index: the order of the rows is kept
		gidx.append(g)
		tables.append(dict(name=names[c], ra=rec[first, 2], dec=rec[first, 3], error=rec[first, 4], area=areas[c]))
	t = orc.nway_match(tables, radius, completeness, prob_ratio_secondary=ratio, densities=densities, scheme=scheme)
	t[own['name']] = t[own['name']] + p_lo
	for c, g in enumerate(gidx):
		col = t[names[c]]
		t[names[c]] = np.where(col >= 0, g[np.maximum(col, 0)] if len(g) else col, -1)
	return t


class OracleSecondarySplitMatch(distributed.SecondarySplitMatch):
	"""the secondary stream split; front half (candidates of the own slices) and back half (table of the
	own primaries from what arrived) are the oracle, the all-to-all-v between them is gloo"""

	def _exchange_device(self):
		return torch.device('cpu')

	def _sync(self):
		pass

	def _build_plan(self):
		self.table = None

	def step(self):
		host = lambda t: t.numpy() if hasattr(t, 'numpy') else np.asarray(t)
		pa = dict(self.primary_all, ra=host(self.primary_all['ra']), dec=host(self.primary_all['dec']), error=host(self.primary_all['error']))
		cands = split_front(pa, self.secondary_slices, self.match_radius, self.scheme)
		received = []
		for c, (p, s_local) in enumerate(cands):
			sl = self.secondary_slices[c]
			p = np.asarray(p, dtype=np.int64)
			s_local = np.asarray(s_local, dtype=np.int64)
			owner = np.searchsorted(self.bounds, p, side='right') - 1
			order = np.argsort(owner, kind='stable')
			rec = np.stack([p[order].astype(float), (s_local[order] + self.sec_offset[c]).astype(float), np.asarray(sl['ra'], dtype=float)[s_local[order]],
				np.asarray(sl['dec'], dtype=float)[s_local[order]],
				np.broadcast_to(np.asarray(sl['error'], dtype=float), np.shape(sl['ra']))[s_local[order]]], axis=1) if len(p) else np.zeros((0, 5))
			send_counts = np.bincount(owner, minlength=self.world).astype(np.int64)
			if self.world > 1:
				sc = torch.as_tensor(send_counts)
				rc = torch.zeros_like(sc)
				dist.all_to_all_single(rc, sc, group=self.group)
				out = torch.zeros((int(rc.sum().item()), 5), dtype=torch.float64)
				dist.all_to_all_single(out, torch.as_tensor(np.ascontiguousarray(rec)), [int(x) for x in rc], [int(x) for x in send_counts], group=self.group)
				got = out.numpy()
			else:
				got = rec
			received.append(got)
		lo = int(self.bounds[self.rank])
		own = dict(self.primary, ra=np.asarray(self.primary['ra'], dtype=float), dec=np.asarray(self.primary['dec'], dtype=float))
		self.table = split_back(own, lo, received, [s['name'] for s in self.secondary_slices], [s['area'] for s in self.secondary_slices],
			self.match_radius, self.prior_completeness, (self.dens, self.dens_plus), self.scheme, self.prob_ratio_secondary)
		return self.table

	def local_rows(self):
		return len(self.table['ncat'])

	def local_table(self):
		return dict(self.table)


class OracleZoneShardedMatch(NumpyMagnitudeHooks, distributed.ZoneShardedMatch):
	"""both sides sharded by declination zones (one all-to-all-v of rows at set-up, through gloo); the match of a zone is the
	numpy oracle with the densities and the cell scheme of the whole catalogues"""

	def _exchange_device(self):
		return torch.device('cpu')

	def _sync(self):
		pass

	def _build_plan(self):
		for z in self.zones:
			z['empty'] = len(z['primary']['ra']) == 0
			z['table'] = None
		self._streams = None
		self.empty = all(z['empty'] for z in self.zones)
		self.table = None

	def _step_zone(self, z, cats, stream):
		import nway_oracle as orc
		tables = [z['primary']] + z['secondaries']
		tables = [dict(t, ra=np.asarray(t['ra']), dec=np.asarray(t['dec']), error=(np.broadcast_to(np.asarray(t['error'], dtype=float), np.shape(t['ra'])))) for t in tables]
		for c in range(1, len(tables)):
			if len(tables[c]['ra']) == 0:
				# (the numpy oracle, like the reference, cannot index an empty catalogue -- the HIP path can, tests/test_hip_parity.py:
				# one source on the far side of the sky stands in; densities and scheme are the whole job's, handed in below)
				p0 = tables[0]
				tables[c] = dict(tables[c], ra=np.array([(np.nanmean(p0['ra']) + 180.0) % 360.0]), dec=np.array([-np.nanmean(p0['dec'])]), error=np.array([1.0]))
		z['table'] = orc.nway_match(tables, self.match_radius, self.prior_completeness, prob_ratio_secondary=self.prob_ratio_secondary,
			densities=(self.dens, self.dens_plus), scheme=self.scheme)

	def step(self):
		for z in self.zones:
			if not z['empty']:
				self._step_zone(z, None, None)
		self.table = self.zones[0].get('table')
		return self.table

	def _empty_table(self):
		names = [self.primary['name']] + [s['name'] for s in self.secondary_slices]
		return dict((n, np.zeros(0, dtype=np.int64)) for n in names + ['ncat', 'match_flag'])

	def local_rows(self):
		return sum(len(z['table']['ncat']) for z in self.zones if not z['empty'])

	def _zone_columns(self, z):
		return dict(z['table']) if not z['empty'] else self._empty_table()
