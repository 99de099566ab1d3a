# development aid (round-5 verdict, "Next round" 6): what the large-table sweep's time is made of on the dense configurations --
# the whole kernel against builds that leave parts of it out (ablations exist in -DNWAYHIP_DEVBUILD builds only; their tables are wrong
# by construction, only their times mean something):   bash tools/dev/sweepbig_split.sh   (on the GPU box; needs the variants below)
#   bash tools/dev/build_variants.sh dev "-DNWAYHIP_DEVBUILD" nolinks "-DNWAYHIP_DEVBUILD -DROUTE_ABLATE=1" noroute "-DNWAYHIP_DEVBUILD -DROUTE_ABLATE=2" \
#        nogather "-DNWAYHIP_DEVBUILD -DROUTE_ABLATE=2 -DSWEEP_BIG_ABLATE=1"
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in c3d c4d "c3s 500000 100000000"; do
	for lib in dev nolinks noroute nogather; do
		echo "== $cfg  lib_$lib"
		NWAYHIP_LIBRARY=$PWD/tools/dev/bin/lib_$lib.so timeout 300 python tools/status_probe.py $cfg 2>&1 | grep -E "^plan|stages us|wall"
	done
done
