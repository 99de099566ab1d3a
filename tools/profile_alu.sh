#!/bin/bash
# The FP64 / issue side of the configurations that no HBM fraction describes (round-5 verdict, "Next round" 3): per configuration a
# kernel trace and ONE pass of SQ counters (never combined with other traces than the kernel trace) of tools/status_probe.py,
# summarised by tools/summarize_alu.py into profiles/alu_<tag>.md.   Run ON THE GPU BOX:
#   gpurun --timeout 1500 -- 'bash tools/profile_alu.sh r06'
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/alu_$TAG
mkdir -p $OUT
cd $ROOT
for cfg in c3s c4s c3d c4d c1x c2x; do
	timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg/trace -- python tools/status_probe.py $cfg > $OUT/$cfg.trace.log 2>&1
	timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/$cfg/sq -- python tools/status_probe.py $cfg > $OUT/$cfg.sq.log 2>&1
done
python tools/summarize_alu.py $OUT > $OUT/alu.md
cat $OUT/alu.md
