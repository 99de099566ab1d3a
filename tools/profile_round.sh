#!/bin/bash
# The profiles committed under profiles/: run ON THE GPU BOX (through gpurun), then summarise here.
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh r01'
#   python tools/summarize_profile.py gpurun_out/r01 profiles r01 && cp gpurun_out/r01/bench.json profiles/bench_r01.json
# Three rocprofv3 passes of the same command (kernel trace + stats; FETCH_SIZE; WRITE_SIZE -- the
# two TCC counters do not fit one pass and --pmc is never combined with other traces) and one
# plain bench run with the CPU baseline.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
CMD="python bench.py --steps 20 --warmup 3 --cpu-sample 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench.json
cat $OUT/bench.json
