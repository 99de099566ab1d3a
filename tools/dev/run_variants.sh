# development aid (GPU box): phase times and bench step of every tools/dev/bin/lib_*.so
for lib in tools/dev/bin/lib_*.so; do
	echo "=== $lib"
	NWAYHIP_LIBRARY=$PWD/$lib timeout 200 python tools/dev/phase_times.py 2>/dev/null | grep -v "^$"
	NWAYHIP_LIBRARY=$PWD/$lib timeout 200 python bench.py --steps 100 --warmup 10 --cpu-sample 0 2>/dev/null | python tools/bench_summary.py
done
