# development aid: instruction-cache counters of the bench's kernels
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u > $OUT/names.txt
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -- python bench.py --steps 4 --warmup 1 --cpu-sample 0 --two-pipelines 0 > $OUT/a.log 2>&1
python - <<PY
import csv, glob
acc = {}
for f in glob.glob('$OUT/a/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:36], r['Counter_Name'])
        acc.setdefault(k, []).append(float(r['Counter_Value']))
for n in sorted(set(k[0] for k in acc)):
    if not n.startswith(('k_', 'void k_')): continue
    print(n)
    for (kn, c), v in sorted(acc.items()):
        if kn == n: print('    %-30s %14.0f' % (c, sum(v) / len(v)))
PY
tail -3 $OUT/a.log
