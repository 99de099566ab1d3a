// libnwayhip: MI355X (gfx950, CDNA4) kernels for nway's match-probability hot path.
//
// Pipeline (one nwayhip_match_enqueue = everything below on one stream, no host sync):
//
//   clear+register  primaries -> cell table (16-byte slots) + fine filter (L2) + coarse filter
//              (LDS copy) + per-primary lon / sin,cos lat                      [front.inc]
//   sweep_c    stream ra/dec of secondary catalogue c once (16 B/lane nontemporal loads),
//              LDS coarse filter, 1 bit of the L2 filter, per-wave LDS staging   [HBM bound]
//   pairs_c    survivors -> cell table -> Vincenty separation -> (p, s, sep) links
//   lists_c    links grouped per primary (scan + scatter)            [scan.inc, lists.inc]
//   k == 2:    finish2 = order + rows + group statistics in one launch     [finish2.inc]
//   k >= 3:    segment order, breadth-first tuple expansion (count -> scan -> fill, rows stay
//              lexicographic), rows kernel, optional correction, groups  [expand.inc, rows.inc]
//
// Semantics follow /root/reference (nway v4.7.1); file:line citations at each kernel.
// Arithmetic is IEEE double with contraction off so that operation order matches the
// reference's numpy expressions (only libm differs: ocml vs glibc, <= 1-2 ulp).

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "nwayhip.h"

#pragma clang fp contract(off)

#include "common.inc"
#include "front.inc"
#include "scan.inc"
#include "lists.inc"
#include "expand.inc"
#include "rows.inc"
#include "finish2.inc"
#include "tail2.inc"
#include "tailk.inc"
#include "elementwise.inc"
#include "api.inc"
#include "plan.inc"
