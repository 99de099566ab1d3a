# development aid: stage times, kernel trace and SQ counters of the bench:  bash tools/dev/prof_sq.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 50 --warmup 5 --cpu-sample 0 --profile-stages "$@" 2> $OUT/stages.err | tail -1 > $OUT/stages.json
python - <<PY
import json
d = json.load(open('$OUT/stages.json'))
print('step %.1f us  ' % (d['ms_per_step'] * 1e3) + '  '.join('%s %.1f' % (k, v * 1e3) for k, v in d['stages_ms'].items() if v))
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 20 --warmup 3 --cpu-sample 0 "$@" > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -- python bench.py --steps 4 --warmup 1 --cpu-sample 0 "$@" > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq2 -- python bench.py --steps 4 --warmup 1 --cpu-sample 0 "$@" > $OUT/sq2.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob('$OUT/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print('%-44s %5s %9.2f us' % (r['Name'].replace('(anonymous namespace)::','').split('(')[0][:44], r['Calls'], float(r['AverageNs'])/1e3))
acc = {}
for f in glob.glob('$OUT/sq*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:36], r['Counter_Name'])
        acc.setdefault(k, []).append(float(r['Counter_Value']))
names = sorted(set(k[0] for k in acc))
for n in names:
    print(n)
    for (kn, c), v in sorted(acc.items()):
        if kn == n: print('    %-22s %14.0f' % (c, sum(v) / len(v)))
PY
