#!/bin/bash
# development aid: registers, spills and scratch of the kernels whose (mangled) name matches $1, from a device-only compile
#   bash tools/dev/regs.sh k_sweep [extra -D flags]
pat=$1; shift
mkdir -p /tmp/isa2
/opt/rocm/lib/llvm/bin/clang++ --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/include -I/root/repo/nway_amd/csrc "$@" -x hip /root/repo/nway_amd/csrc/nwayhip.hip --cuda-device-only -S -o /tmp/isa2/dev.s 2>/dev/null
python - "$pat" <<'PY'
import re, sys
txt = open('/tmp/isa2/dev.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)', txt):
	if sys.argv[1] in m.group(1):
		print('%-90s scratch %4s sgpr %3s (spilled %3s) vgpr %3s (spilled %3s)' % (m.group(1)[:90], m.group(2), m.group(3), m.group(4), m.group(5), m.group(6)))
PY
