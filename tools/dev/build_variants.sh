# development aid: build variants of the library with extra -D flags into tools/dev/bin/lib_<name>.so
#   bash tools/dev/build_variants.sh name1 "-DA=1 -DB=2" name2 "-DC=3" ...
mkdir -p tools/dev/bin
while [ $# -ge 2 ]; do
	name=$1; flags=$2; shift 2
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Inway_amd/csrc -Wno-unused-function $flags nway_amd/csrc/nwayhip.hip -o tools/dev/bin/lib_$name.so &
done
wait
ls -la tools/dev/bin/
