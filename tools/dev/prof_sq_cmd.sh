# development aid: SQ counters (two passes) of the kernels of any command:  bash tools/dev/prof_sq_cmd.sh <tag> <command ...>
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -- "$@" > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq2 -- "$@" > $OUT/sq2.log 2>&1
python - <<PY
import csv, glob
acc = {}
for f in glob.glob('$OUT/sq*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:36], r['Counter_Name'])
        acc.setdefault(k, []).append(float(r['Counter_Value']))
names = sorted(set(k[0] for k in acc))
for n in names:
    if not n.replace('void ', '').startswith('k_'): continue
    print(n)
    for (kn, c), v in sorted(acc.items()):
        if kn == n: print('    %-22s %14.0f' % (c, sum(v) / len(v)))
PY
