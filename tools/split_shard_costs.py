"""What ONE rank of an N-GPU secondary-split job costs per step, measured on one GPU: the rank
registers all primaries, sweeps its 1/N slice of the secondaries and exports the candidates
(front half), and imports + finishes the rows of its n0/N own primaries (back half).  The
all-to-all between the halves needs the other GPUs and is NOT measured here (its payload is
printed); the back half here sees only the candidates this rank found itself (1/N of what a real
run delivers), i.e. its time is a slight underestimate.

    python tools/split_shard_costs.py [n_primary] [n_secondary] [radius]      (on the GPU box)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nway_amd  # noqa: E402
from nway_amd import _hip, distributed  # noqa: E402

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
radius = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
dev = torch.device('cuda', 0)
primary, secondary = bench.make_workload(n0, n1, 1)
log = nway_amd.NullOutputLogger()
err = radius / 3600.
dens, dens_plus = nway_amd._densities_from_sizes(['P', 'S'], [n0, n1], [bench.SKY_AREA] * 2, log)
comp = nway_amd._completeness_vector(0.9, 2)
print('| GPUs N | front half: register %d primaries + sweep of n1/N secondaries incl. export (us, stage events) | all-to-all payload per rank | back half (us, stage events) | both halves back to back, wall clock, no exchange (us) |' % n0)
print('|---|---|---|---|---|')
for world in (1, 2, 4, 8):
	pb = distributed.shard_bounds(n0, world)
	sb = distributed.shard_bounds(n1, world)
	params = _hip.make_params(2, _hip.SCHEME_SPHERE, radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp))
	cats = [_hip.DeviceCatalogue(primary['ra'], primary['dec'], primary['error'], dev),
		_hip.DeviceCatalogue(secondary['ra'][sb[0]:sb[1]], secondary['dec'][sb[0]:sb[1]], 0.1, dev)]
	_, cap_rows = nway_amd._estimate_capacities([int(pb[1])] + [n1], [bench.SKY_AREA] * 2, radius, _hip.SCHEME_SPHERE, True)
	plan = _hip.MatchPlan([c.n for c in cats], params, 65536, cap_rows, dev, lean=True)
	assert plan.fused
	capacity = max(1024, 4 * int(pb[1]) // world + 1024)
	nbytes = plan.split_buffer_bytes(world, capacity)
	export = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
	bounds = torch.as_tensor(pb).to(dev)
	sp = _hip.Split()
	sp.world, sp.rank = world, 0
	sp.d_bounds = bounds.data_ptr()
	sp.h_p_lo, sp.h_p_hi = int(pb[0]), int(pb[1])
	sp.slice_offset[1] = 0
	sp.capacity = capacity
	sp.d_export = export.data_ptr()
	sp.d_import = export.data_ptr()  # (this rank's own block 0; the other blocks are for other owners and are skipped)

	def run(which, reps):
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(reps):
			if which in ('front', 'both'):
				plan.split_front(cats, sp)
			if which in ('back', 'both'):
				plan.split_back(cats, sp)
		torch.cuda.synchronize()
		return (time.perf_counter() - t0) / reps * 1e6
	run('both', 5)
	both = run('both', 40)
	# the halves, from stage events on the stream (register + sweep | tail; the import launch sits between them)
	plan.profile(0xff)
	for _ in range(20):
		plan.split_front(cats, sp)
		plan.split_back(cats, sp)
	torch.cuda.synchronize()
	nl, ms = plan.profile_read()
	plan.profile(0)
	stage = dict((nm, 1e3 * m / 20) for nm, m in zip(_hip.STAGE_NAMES, ms))
	heads = export.view(torch.int32)[::(capacity + 1) * 12][:world].cpu().numpy()  # (zeroed by the back half: read the status instead)
	st = plan.read_status()
	assert int(st[_hip.ST_FLAGS]) == 0, st[:4]
	links = int(st[_hip.ST_TESTS])
	print('| %d | register %.1f + sweep %.1f | ~%d records = %.0f KB | tail %.1f (+ import) | %.1f |' % (
		world, stage['register'], stage['sweep'], links, links * 48 / 1e3, stage['rows'], both))
	plan.close()

# ---- what the host round trip between the two halves costs (the collectives are torch.distributed calls, not part of
# the C ABI): ONE rank -- SecondarySplitMatch.step() = front half, exchange (a device copy at world 1; the all-to-all
# otherwise), back half, each enqueued from Python -- against the unsplit pipeline enqueued with one call
eng = distributed.SecondarySplitMatch(primary, [dict(secondary, error=0.1)], radius, 0.9, dev)
for _ in range(20):
	eng.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
	eng.step()
torch.cuda.synchronize()
split_us = (time.perf_counter() - t0) / 100 * 1e6
t0 = time.perf_counter()
for _ in range(100):
	eng.plan.split_front(eng.cats, eng.split)
	eng.plan.split_back(eng.cats, eng.split)
torch.cuda.synchronize()
halves_us = (time.perf_counter() - t0) / 100 * 1e6
params = _hip.make_params(2, _hip.SCHEME_SPHERE, radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp))
cats = [_hip.DeviceCatalogue(primary['ra'], primary['dec'], primary['error'], dev), _hip.DeviceCatalogue(secondary['ra'], secondary['dec'], 0.1, dev)]
cap_pairs, cap_rows = nway_amd._estimate_capacities([n0, n1], [bench.SKY_AREA] * 2, radius, _hip.SCHEME_SPHERE, True)
plan, _ = _hip.run_plan([n0, n1], params, cats, cap_pairs, cap_rows, dev, lean=True)
for _ in range(20):
	plan.enqueue(cats)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
	plan.enqueue(cats)
torch.cuda.synchronize()
whole_us = (time.perf_counter() - t0) / 100 * 1e6
print()
print('one rank, %d x %d, per step (wall clock, 100 steps, one secondary buffer): unsplit pipeline, one enqueue %.1f us | both halves back to back '
	'(two enqueues, no exchange) %.1f us | SecondarySplitMatch.step(): front, device copy of the %d-byte export buffer, back %.1f us' % (
	n0, n1, whole_us, halves_us, eng.export.numel(), split_us))
