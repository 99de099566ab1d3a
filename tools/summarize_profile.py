"""Condense rocprofv3 outputs (kernel_stats.csv, counter_collection.csv of separate FETCH_SIZE /
WRITE_SIZE passes) into the small text/JSON summaries committed under profiles/.

    python tools/summarize_profile.py gpurun_out/r01 profiles r01 [n_secondary]
"""
import csv
import glob
import json
import os
import sys

import hashlib
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
	"""sha256 over the kernel sources (nway_amd/csrc/*.inc, *.hip, include/nwayhip.h), in name order: what bench.py compares with the
	tree it runs in, so that a counter summary of an OLDER build is recognised as such"""
	h = hashlib.sha256()
	files = sorted(glob.glob(os.path.join(ROOT, 'nway_amd', 'csrc', '*.inc')) + glob.glob(os.path.join(ROOT, 'nway_amd', 'csrc', '*.hip'))) + [
		os.path.join(ROOT, 'include', 'nwayhip.h')]
	for f in files:
		h.update(os.path.basename(f).encode())
		h.update(open(f, 'rb').read())
	return h.hexdigest()[:16]


src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
n_secondary = int(sys.argv[4]) if len(sys.argv) > 4 else 10000000
os.makedirs(dst, exist_ok=True)


def short(name):
	name = name.replace('(anonymous namespace)::', '')
	return name.split('(')[0].replace('void ', '')


lines = []
stats = sorted(glob.glob(os.path.join(src, 'stats', '*', '*kernel_stats.csv')), key=os.path.getmtime, reverse=True)  # (the newest run of the directory)
kernel_avg = {}
if stats:
	lines.append('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --cpu-sample 0')
	lines.append('%-34s %6s %12s %10s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'max_us', 'pct'))
	for r in csv.DictReader(open(stats[0])):
		nm = short(r['Name'])
		kernel_avg[nm] = float(r['AverageNs']) / 1e3
		lines.append('%-34s %6s %12.1f %10.2f %10.2f %7s' % (nm[:34], r['Calls'], float(r['TotalDurationNs']) / 1e3,
			float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage']))
counters = {}
for which in ('fetch', 'write'):
	files = sorted(glob.glob(os.path.join(src, which, '*', '*counter_collection.csv')), key=os.path.getmtime, reverse=True)
	if not files:
		continue
	per_kernel = {}
	for r in csv.DictReader(open(files[0])):
		per_kernel.setdefault((short(r['Kernel_Name']), r['Counter_Name']), []).append(float(r['Counter_Value']))
	for (k, c), v in sorted(per_kernel.items()):
		counters[(k, c)] = sum(v) / len(v)
if counters:
	lines.append('')
	lines.append('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per dispatch, KiB as reported')
	for (k, c), v in sorted(counters.items()):
		if k.startswith('k_'):
			lines.append('%-34s %-12s %14.1f' % (k[:34], c, v))
open(os.path.join(dst, 'rocprof_%s.txt' % tag), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

# HBM traffic of the sweep per launch, corrected as MI355X_MICROARCH.md prescribes: bytes = KiB * 1024;
# on gfx950 FETCH_SIZE reports exactly 1/2 of a wide (16 B/lane) coalesced streaming read.  The sweep
# of the sparse path also gathers (tag words, table entries, the survivors' coordinates again, the
# occupancy bitmap once per workgroup): those narrow accesses are counted in full, so only the
# STREAM's uncounted half is added back -- fetch = raw + (16 B x n_secondary) / 2.  WRITE_SIZE as reported.
sweep = [k for (k, c) in counters if k.startswith('k_sweep')]
if sweep:
	k = sweep[0]
	raw = counters.get((k, 'FETCH_SIZE'), 0.0) * 1024
	stream = 16.0 * n_secondary
	fetch = raw + stream / 2 if raw >= stream / 2 else 2 * raw
	write = counters.get((k, 'WRITE_SIZE'), 0.0) * 1024
	rec = dict(round=tag, kernel=k, n_secondary=n_secondary, fetch_size_kib_raw=counters.get((k, 'FETCH_SIZE')),
		write_size_kib_raw=counters.get((k, 'WRITE_SIZE')), hbm_bytes_per_launch=fetch + write,
		algorithmic_bytes_per_launch=stream, avg_launch_us=kernel_avg.get(k),
		correction='FETCH_SIZE KiB*1024 + half of the 16 B/lane coalesced stream (gfx950 reports that half only; the gathers of the '
			'kernel are counted in full), WRITE_SIZE KiB*1024')
	# the build these counters belong to (run this script in the tree that was measured, before editing the kernels)
	rec['kernel_source_sha16'] = kernel_source_hash()
	try:
		rec['git_head'] = subprocess.check_output(['git', '-C', ROOT, 'rev-parse', '--short=12', 'HEAD'], universal_newlines=True).strip()
		rec['git_dirty'] = bool(subprocess.check_output(['git', '-C', ROOT, 'status', '--porcelain', '--', 'nway_amd/csrc', 'include'], universal_newlines=True).strip())
	except Exception:
		rec['git_head'], rec['git_dirty'] = None, None
	json.dump(rec, open(os.path.join(dst, 'sweep_traffic.json'), 'w'), indent=1)
	print(json.dumps(rec))
