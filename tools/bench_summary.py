"""print the interesting numbers of bench.py's JSON line (stdin)"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%s value=%.4g rows/s  ms/step=%.4f  sweep=%.1f us (%.0f GB/s, frac %.3f)  rows=%d surv=%d tests=%d' % (
	' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r['launch_ms'] * 1e3, r['achieved'], r['frac'],
	d['config']['rows_per_step'], d['config']['survivors_per_step_rank0'], d['config']['distance_tests_per_step_rank0']))
if 'stages_ms' in d:
	print('   stages us: ' + '  '.join('%s=%.1f' % (k, v * 1e3) for k, v in d['stages_ms'].items()))
