# FETCH_SIZE / L2 hit counters of the dense 2-way pass (development aid)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_dense
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python tools/status_probe.py c3d > $OUT/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python tools/status_probe.py c3d > $OUT/w.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/h -- python tools/status_probe.py c3d > $OUT/h.log 2>&1
python - <<PY
import csv, glob, collections
for d in ('f','w','h'):
    fs = glob.glob('$OUT/%s/*/*counter_collection.csv' % d)
    if not fs: print(d, 'no file'); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')
        acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(acc.items()):
        if k.startswith('k_'): print('%-28s %-14s %14.1f' % (k[:28], c, sum(v)/len(v)))
PY
