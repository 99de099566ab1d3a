// libnwayhip: MI355X (gfx950, CDNA4) kernels for nway's match-probability hot path.
//
// Pipeline (one nwayhip_match_enqueue = everything below on one stream, no host sync):
//
// SPARSE inputs (every secondary catalogue expects < 0.5 chance neighbours per primary), 3 launches:
//   register   primaries -> direct-mapped cell table: ONE atomic per cell (the occupancy bit is
//              the slot claim), 64-byte record + tag byte by plain stores          [sparse.inc]
//   sweep      stream ra/dec of the secondary catalogue(s) once (16 B per lane per column),
//              occupancy bitmap in LDS, one tag word from L2; every workgroup then probes its
//              own survivors: record line -> Vincenty -> link in the primary's slots
//                                                              [front.inc, HBM bound]
//   tail       single-pass scan over the primaries' row counts + rows + group statistics
//                                                              [tail2.inc, tailk.inc]
// GENERAL path:
//   clear+register  primaries -> cell table (16-byte slots) + fine filter (L2) + coarse filter
//              (LDS copy) + per-primary lon / sin,cos lat                      [front.inc]
//   sweep_c    the same streaming kernel, survivors into per-workgroup regions
//   pairs_c    survivors -> cell table -> Vincenty separation -> (p, s, sep) links
//   lists_c    links grouped per primary (scan + scatter)            [scan.inc, lists.inc]
//   k == 2:    rows (row-parallel) + group statistics                     [finish2.inc]
//   k >= 3:    segment order, breadth-first tuple expansion (count -> scan -> fill, rows stay
//              lexicographic), rows kernel, optional correction, groups  [expand.inc, rows.inc]
//
// Semantics follow /root/reference (nway v4.7.1); file:line citations at each kernel.
// Arithmetic is IEEE double with contraction off so that operation order matches the
// reference's numpy expressions (only libm differs: ocml vs glibc, <= 1-2 ulp).

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

#include "nwayhip.h"

#ifdef NWAYHIP_DEVBUILD
// development builds only: with NWAYHIP_LAUNCH_LOG=<file> every launch is followed by a synchronisation and a pause, and the file
// always names the launch in flight and the one before it -- what a device fault that the runtime reports without a kernel
// name (a posted store outside its buffer) is traced with (tools/dev/fault_loop.sh)
#include <fcntl.h>
#include <unistd.h>
static int dev_launch_fd() {
	static int fd = -2;
	if (fd == -2) {
		const char* e = getenv("NWAYHIP_LAUNCH_LOG");
		fd = (e && *e) ? open(e, O_WRONLY | O_CREAT | O_TRUNC, 0644) : -1;
	}
	return fd;
}
static void dev_launch_note(const char* name, int state, int line) {
	static char prev[200] = "";
	static long long count = 0;
	const int fd = dev_launch_fd();
	if (fd < 0) return;
	char buf[512];
	memset(buf, ' ', sizeof(buf));
	const int n = snprintf(buf, sizeof(buf), "#%lld %s %s (plan.inc:%d) | before: %s", ++count, state ? "finished" : "IN FLIGHT", name, line, prev);
	if (n > 0 && n < (int)sizeof(buf)) buf[n] = ' ';
	buf[sizeof(buf) - 1] = '\n';
	if (pwrite(fd, buf, sizeof(buf), 0) < 0) return;
	if (state) snprintf(prev, sizeof(prev), "%s:%d", name, line);
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                    \
	do {                                                                                                  \
		dev_launch_note(#kernelName, 0, __LINE__);                                                        \
		hipLaunchKernelGGLInternal((kernelName), (numBlocks), (numThreads), (memPerBlock), (streamId), __VA_ARGS__); \
		if (dev_launch_fd() >= 0) {                                                                       \
			(void)hipStreamSynchronize(streamId);                                                         \
			usleep(1500);                                                                                 \
			dev_launch_note(#kernelName, 1, __LINE__);                                                    \
		}                                                                                                 \
	} while (0)
#endif

#pragma clang fp contract(off)

#include "fastmath.inc"
#include "common.inc"
#include "sparse.inc"
#include "front.inc"
#include "sweepbig.inc"
#include "scan.inc"
#include "lists.inc"
#include "expand.inc"
#include "rows.inc"
#include "finish2.inc"
#include "tail2.inc"
#include "zones.inc"
#include "taild.inc"
#include "tailk.inc"
#include "tail3q.inc"
#include "taild3.inc"
#include "elementwise.inc"
#include "api.inc"
#include "plan.inc"
#include "comm.inc"
