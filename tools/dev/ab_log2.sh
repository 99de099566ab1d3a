# development aid: the headline pass with the cell table at 2^18 / 2^19 / 2^20 positions (NWAYHIP_DEV=1 NWAYHIP_DIRECT_LOG2)
for rep in 1 2; do for lg in 20 19 18; do
  NWAYHIP_DEV=1 NWAYHIP_DIRECT_LOG2=$lg timeout 200 python bench.py --steps 50 --warmup 5 --cpu-sample 0 --extras 0 --fixed-jobs 0 --two-pipelines 0 --live-traffic 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('log2 $lg: %.2f us per pass, sweep %.2f us, survivors %d, registrations %d, rows %d, check %s' % (d['ms_per_step'] * 1e3, d['roofline']['launch_ms'] * 1e3, d['config']['survivors_per_step_rank0'], d['config']['registrations_rank0'], d['config']['rows_per_step'], d.get('check', {}).get('ok')))"
done; done
