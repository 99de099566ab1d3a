#!/bin/bash
# The profiles committed under profiles/: run ON THE GPU BOX (through gpurun), then summarise here.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r03'
#   python tools/summarize_profile.py gpurun_out/r03 profiles r03 && cp gpurun_out/r03/bench.json profiles/bench_r03.json
#   cp gpurun_out/r03/configs.md profiles/configs_r03.md; cp gpurun_out/r03/api_timing.md profiles/api_timing_r03.md
# Three rocprofv3 passes of the same command (kernel trace + stats; FETCH_SIZE; WRITE_SIZE -- the
# two TCC counters do not fit one pass and --pmc is never combined with other traces), one
# plain bench run with the CPU baseline, the table of all configurations and the API timings.
TAG=${1:-r03}
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
timeout 900 python tools/config_table.py > $OUT/configs.md 2> $OUT/configs.err
cat $OUT/configs.md
timeout 600 python tools/api_timing.py > $OUT/api_timing.md 2> $OUT/api_timing.err
cat $OUT/api_timing.md
