# development aid (GPU box): bench step (and, for -DNWAYHIP_DEVBUILD builds, phase times) of every tools/dev/bin/lib_*.so
#   bash tools/dev/run_variants.sh [phase]
for lib in nway_amd/csrc/libnwayhip.so tools/dev/bin/lib_*.so; do
	echo "=== $lib"
	if [ "$1" = phase ] && [ "$lib" != nway_amd/csrc/libnwayhip.so ]; then
		NWAYHIP_LIBRARY=$PWD/$lib timeout 200 python tools/dev/phase_times.py 2>/dev/null | grep -v "^$"
	fi
	for rep in 1 2; do
		NWAYHIP_LIBRARY=$PWD/$lib timeout 200 python bench.py --steps 100 --warmup 10 --cpu-sample 0 --two-pipelines 0 2>/dev/null | python tools/bench_summary.py
	done
done
