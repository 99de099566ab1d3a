// libnwayhip: MI355X (gfx950, CDNA4) kernels for nway's match-probability hot path.
//
// Pipeline (one nwayhip_match_enqueue = everything below on one stream, no host sync):
//
//   register   primaries -> cell hash + bitmap (+ per-primary lon / sin,cos lat)
//   sweep_c    stream ra/dec of secondary catalogue c once (coalesced 16 B/lane),
//              1 bitmap probe per source, ballot-compact the survivors      [HBM bound]
//   pairs_c    survivors -> hash lookup -> Vincenty separation -> (p, s, sep) links
//   lists_c    per-primary neighbour lists L_c(p), ascending secondary index
//   expand     breadth-first tuple expansion, one catalogue per level, rows stay
//              lexicographically ordered (count -> scan -> fill)
//   rows       separations, Separation_max, ncat, log_bf, prior, dist_post
//   groups     per-primary log-sum-exp, p_any, p_i, match_flag (wave64 shuffles)
//
// Semantics follow /root/reference (nway v4.7.1); file:line citations at each kernel.
// Arithmetic is IEEE double with contraction off so that operation order matches the
// reference's numpy expressions (only libm differs: ocml vs glibc, <= 1-2 ulp).

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "nwayhip.h"

#pragma clang fp contract(off)

namespace {

constexpr int BLOCK = 256;
constexpr int WAVE = 64;
constexpr int SCAN_BLOCKS = 512;       // chunks of the device-side-length scans
constexpr int MAXREG_SPHERE = 64;      // cap on cell registrations of one primary (all-sky scheme)
constexpr unsigned long long EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;
typedef double dbl2 __attribute__((ext_vector_type(2)));

thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return -1;
}

#define HIP_TRY(expr)                                                                  \
	do {                                                                               \
		hipError_t e_ = (expr);                                                        \
		if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_)); \
	} while (0)

// ------------------------------------------------------------------------------------
// device-side parameter blocks
// ------------------------------------------------------------------------------------

struct CellParams {
	int scheme;
	double err_deg;        // flat: cell edge (fastskymatch.py:125)
	// all-sky scheme: declination bands of height h, RA bins tapering towards the poles
	double inv_h;          // 1 / band height (deg^-1)
	int nbands;
	int nra_eq;            // RA bins at the equator
	int taper;             // bands (from a pole) over which the bin count grows linearly
	double reach_deg;      // radius inflated by rounding slack, degrees
	// flat-cell applicability test (fastskymatch.py:96)
	double ra_lo, ra_hi;   // 10*err, 360 - 10*err
};

struct HashTable {
	unsigned long long* keys;
	int32_t* vals;
	uint32_t slot_mask;
	uint32_t* bitmap;
	uint32_t bit_mask;     // nbits - 1
};

struct CatView {
	const double* ra;
	const double* dec;
	const double* sigma;
	double sigma_const;
	int64_t n;
};

struct Cats {
	CatView c[NWAYHIP_MAXCAT];
};

// ------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t hash_bits(uint32_t i, uint32_t j) {
	uint32_t h = (i * 0x9E3779B1u) ^ (j * 0x85EBCA77u);
	h ^= h >> 15;
	h *= 0x2C1B3C6Du;
	h ^= h >> 12;
	h *= 0x297A2D39u;
	h ^= h >> 15;
	return h;
}

__device__ __forceinline__ uint32_t hash_slot(uint32_t i, uint32_t j) {
	uint32_t h = (i * 0xC2B2AE3Du) + (j * 0x27D4EB2Fu);
	h ^= h >> 16;
	h *= 0x165667B1u;
	h ^= h >> 13;
	return h;
}

__device__ __forceinline__ unsigned long long pack_key(int32_t i, int32_t j) {
	return ((unsigned long long)(uint32_t)i << 32) | (unsigned long long)(uint32_t)j;
}

// Flat cells: i, j = int(ra / err), int(dec / err) -- true division, truncation toward
// zero (fastskymatch.py:125).
__device__ __forceinline__ void flat_cell(double ra, double dec, double err, int32_t& i, int32_t& j) {
	i = (int32_t)(long long)(ra / err);
	j = (int32_t)(long long)(dec / err);
}

// All-sky cells (our own scheme; the reference's HEALPix buckets are not reproduced, only
// their post-filter result).  Band j = floor((dec + 90) / h); the number of RA bins of a
// band grows linearly with its distance from the nearer pole and saturates at nra_eq.
__device__ __forceinline__ int sphere_band(double dec, const CellParams& cp) {
	int j = (int)((dec + 90.0) * cp.inv_h);
	return min(max(j, 0), cp.nbands - 1);
}

__device__ __forceinline__ int sphere_nra(int j, const CellParams& cp) {
	int t = min(j, cp.nbands - 1 - j);
	if (t >= cp.taper) return cp.nra_eq;
	long long n = ((long long)cp.nra_eq * (long long)t) / (long long)cp.taper;
	return (int)max(n, 1ll);
}

__device__ __forceinline__ int sphere_bin(double ra, int nra) {
	double x = ra * (1.0 / 360.0);
	x -= floor(x);
	int i = (int)(x * (double)nra);
	return min(i, nra - 1);
}

__device__ __forceinline__ void sphere_cell(double ra, double dec, const CellParams& cp, int32_t& i, int32_t& j) {
	j = sphere_band(dec, cp);
	i = sphere_bin(ra, sphere_nra(j, cp));
}

// condition of fastskymatch.py:96 for one source (err < 1 is checked on the host)
__device__ __forceinline__ bool flat_condition(double ra, double dec, const CellParams& cp) {
	return (ra > cp.ra_lo) && (ra < cp.ra_hi) && (fabs(dec) < 45.0);
}

// per-source quantities of dist() (fastskymatch.py:32-41)
struct SkyPoint {
	double lon, slat, clat;
};

__device__ __forceinline__ SkyPoint sky_point(double ra, double dec) {
	SkyPoint s;
	s.lon = ra / 180 * M_PI;
	double lat = dec / 180 * M_PI;
	sincos(lat, &s.slat, &s.clat);
	return s;
}

// fastskymatch.py:36-47, then "* 60 * 60" (__init__.py:163): separation in arcsec.
// a = the earlier catalogue, b = the later one, as in __init__.py:152.
__device__ __forceinline__ double separation_arcsec(const SkyPoint& a, const SkyPoint& b) {
	double dlon = b.lon - a.lon;
	double sdlon, cdlon;
	sincos(dlon, &sdlon, &cdlon);
	double num1 = b.clat * sdlon;
	double num2 = a.clat * b.slat - a.slat * b.clat * cdlon;
	double den = a.slat * b.slat + a.clat * b.clat * cdlon;
	double deg = atan2(hypot(num1, num2), den) * 180 / M_PI;
	return deg * 60 * 60;
}

__device__ __forceinline__ double exp10_ref(double x) {
	return pow(10.0, x);  // numpy: 10**x
}

// bayesdistance.py:26-32
__device__ __forceinline__ double posterior_ref(double prior, double logbf) {
	return 1. / (1 + (1 - prior) * exp10_ref(-logbf - log10(prior)));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
	for (int o = WAVE / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
	return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
	for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

// exclusive scan of one int per thread over a 256-thread block; returns the block total in `total`
__device__ __forceinline__ long long block_exclusive_scan(long long v, long long& total) {
	__shared__ long long wave_tot[BLOCK / WAVE];
	int lane = lane_id();
	int wid = threadIdx.x / WAVE;
	long long incl = v;
#pragma unroll
	for (int o = 1; o < WAVE; o <<= 1) {
		long long t = __shfl_up(incl, o);
		if (lane >= o) incl += t;
	}
	__syncthreads();  // protect wave_tot from a previous call
	if (lane == WAVE - 1) wave_tot[wid] = incl;
	__syncthreads();
	long long base = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < BLOCK / WAVE; ++w) {
		long long x = wave_tot[w];
		if (w < wid) base += x;
		tot += x;
	}
	total = tot;
	return base + incl - v;
}

__device__ __forceinline__ long long chunk_of(long long n, int nblocks) {
	long long c = (n + nblocks - 1) / nblocks;
	return (c + BLOCK - 1) / BLOCK * BLOCK;
}

// ------------------------------------------------------------------------------------
// K1/K2  primary registration                       fastskymatch.py:118-133 ("only the
//        primary catalogue is allowed to define new buckets")
// ------------------------------------------------------------------------------------

__device__ __forceinline__ bool hash_insert(const HashTable& ht, int32_t ci, int32_t cj, int32_t p) {
	uint32_t hb = hash_bits((uint32_t)ci, (uint32_t)cj) & ht.bit_mask;
	atomicOr(&ht.bitmap[hb >> 5], 1u << (hb & 31));
	unsigned long long key = pack_key(ci, cj);
	uint32_t slot = hash_slot((uint32_t)ci, (uint32_t)cj) & ht.slot_mask;
	for (uint32_t probe = 0; probe <= ht.slot_mask; ++probe) {
		unsigned long long old = atomicCAS(&ht.keys[slot], EMPTY_KEY, key);
		if (old == EMPTY_KEY) {
			ht.vals[slot] = p;
			return true;
		}
		slot = (slot + 1) & ht.slot_mask;
	}
	return false;
}

__global__ void __launch_bounds__(BLOCK) k_register(CatView prim, CellParams cp, HashTable ht, double* plon,
	double* pslat, double* pclat, long long* status) {
	long long p = (long long)blockIdx.x * BLOCK + threadIdx.x;
	if (p >= prim.n) return;
	double ra = prim.ra[p], dec = prim.dec[p];
	SkyPoint sp = sky_point(ra, dec);
	plon[p] = sp.lon;
	pslat[p] = sp.slat;
	pclat[p] = sp.clat;
	if (!flat_condition(ra, dec, cp)) status[NWAYHIP_ST_NOTFLAT + 0] = 1;
	if (!(ra == ra) || !(dec == dec)) return;  // NaN coordinates never match
	int nreg = 0;
	bool ok = true;
	if (cp.scheme == NWAYHIP_SCHEME_FLAT) {
		// A source in cell (i, j) sits in buckets (i..i+1, j..j+1) (:128); two sources share a
		// bucket iff their cells differ by <= 1 in both axes.  Registering the primary in its
		// 3x3 neighbourhood lets every secondary probe only its own cell.
		int32_t i, j;
		flat_cell(ra, dec, cp.err_deg, i, j);
		for (int di = -1; di <= 1; ++di)
			for (int dj = -1; dj <= 1; ++dj) {
				ok &= hash_insert(ht, i + di, j + dj, (int32_t)p);
				++nreg;
			}
	} else {
		// every cell that a point within `reach` of the primary can fall into
		double reach = cp.reach_deg;
		int jlo = sphere_band(dec - reach, cp), jhi = sphere_band(dec + reach, cp);
		double half;  // largest RA offset of a point within reach (degrees); 180 = whole circle
		if (fabs(dec) + reach >= 90.0) {
			half = 180.0;
		} else {
			double s = sin(reach / 180 * M_PI) / cos(dec / 180 * M_PI);
			half = (s >= 1.0) ? 180.0 : asin(s) * (180.0 / M_PI) * (1 + 1e-9) + 1e-12;
		}
		for (int j = jlo; j <= jhi && ok; ++j) {
			int nra = sphere_nra(j, cp);
			if (2 * half >= 360.0 - 360.0 / nra) {
				for (int i = 0; i < nra && ok; ++i) {
					if (++nreg > MAXREG_SPHERE) { ok = false; break; }
					ok &= hash_insert(ht, i, j, (int32_t)p);
				}
			} else {
				int ilo = sphere_bin(ra - half, nra), ihi = sphere_bin(ra + half, nra);
				for (int i = ilo;; i = (i + 1 == nra) ? 0 : i + 1) {
					if (++nreg > MAXREG_SPHERE) { ok = false; break; }
					ok &= hash_insert(ht, i, j, (int32_t)p);
					if (i == ihi) break;
				}
			}
		}
	}
	if (!ok) atomicOr((unsigned long long*)&status[NWAYHIP_ST_FLAGS], (unsigned long long)NWAYHIP_FLAG_REG_OVERFLOW);
	atomicAdd((unsigned long long*)&status[NWAYHIP_ST_REGISTRATIONS], (unsigned long long)nreg);
}

// ------------------------------------------------------------------------------------
// K3a  sweep: stream one secondary catalogue, keep sources whose cell is registered
//      (fastskymatch.py:122-133 for ti > 0: "if k in buckets").  HBM-bound: 16 B per source.
// ------------------------------------------------------------------------------------

template <int SCHEME>
__device__ __forceinline__ bool sweep_probe(double ra, double dec, const CellParams& cp, const HashTable& ht) {
	int32_t i, j;
	if (SCHEME == NWAYHIP_SCHEME_FLAT)
		flat_cell(ra, dec, cp.err_deg, i, j);
	else
		sphere_cell(ra, dec, cp, i, j);
	uint32_t hb = hash_bits((uint32_t)i, (uint32_t)j) & ht.bit_mask;
	uint32_t word = ht.bitmap[hb >> 5];
	return ((word >> (hb & 31)) & 1u) && (ra == ra) && (dec == dec);
}

__device__ __forceinline__ void wave_append(bool pass, int32_t value, int32_t* out, long long* counter) {
	unsigned long long mask = __ballot(pass);
	if (mask == 0) return;
	int lane = lane_id();
	long long base = 0;
	if (lane == 0) base = (long long)atomicAdd((unsigned long long*)counter, (unsigned long long)__popcll(mask));
	base = __shfl(base, 0);
	if (pass) out[base + __popcll(mask & ((1ull << lane) - 1ull))] = value;
}

template <int SCHEME>
__global__ void __launch_bounds__(BLOCK) k_sweep(const double* __restrict__ ra, const double* __restrict__ dec,
	long long n, CellParams cp, HashTable ht, int32_t* __restrict__ surv, long long* n_surv, long long* notflat) {
	const long long npair = n >> 1;
	const long long stride = (long long)gridDim.x * BLOCK;
	bool bad = false;
	const dbl2* ra2 = reinterpret_cast<const dbl2*>(ra);
	const dbl2* dec2 = reinterpret_cast<const dbl2*>(dec);
	for (long long v = (long long)blockIdx.x * BLOCK + threadIdx.x; v < ((npair + WAVE - 1) / WAVE) * WAVE; v += stride) {
		bool in = v < npair;
		dbl2 r = {0.0, 0.0}, d = {0.0, 0.0};
		if (in) {
			r = __builtin_nontemporal_load(&ra2[v]);
			d = __builtin_nontemporal_load(&dec2[v]);
		}
		bool p0 = in && sweep_probe<SCHEME>(r.x, d.x, cp, ht);
		bool p1 = in && sweep_probe<SCHEME>(r.y, d.y, cp, ht);
		if (in) bad |= !(flat_condition(r.x, d.x, cp) && flat_condition(r.y, d.y, cp));
		wave_append(p0, (int32_t)(2 * v), surv, n_surv);
		wave_append(p1, (int32_t)(2 * v + 1), surv, n_surv);
	}
	if ((n & 1) && blockIdx.x == 0 && threadIdx.x < WAVE) {
		bool in = threadIdx.x == 0;
		double r = in ? ra[n - 1] : 0, d = in ? dec[n - 1] : 0;
		bool p0 = in && sweep_probe<SCHEME>(r, d, cp, ht);
		if (in) bad |= !flat_condition(r, d, cp);
		wave_append(p0, (int32_t)(n - 1), surv, n_surv);
	}
	if (__any(bad) && lane_id() == 0) *notflat = 1;
}

// ------------------------------------------------------------------------------------
// K3b  pairs: survivors -> registered primaries of the cell -> separation -> links
//      (the per-bucket product of fastskymatch.py:174-181 restricted to one secondary
//       catalogue, plus the radius filter of __init__.py:152-166,180)
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BLOCK) k_pairs(CatView sec, CellParams cp, HashTable ht, const double* __restrict__ plon,
	const double* __restrict__ pslat, const double* __restrict__ pclat, const int32_t* __restrict__ surv,
	const long long* n_surv_ptr, int radius_filter, double radius_arcsec, int32_t* pair_p, int32_t* pair_s,
	double* pair_sep, long long cap_pairs, int32_t* cnt, long long* n_pairs, long long* status) {
	const long long n_surv = *n_surv_ptr;
	const long long stride = (long long)gridDim.x * BLOCK;
	unsigned long long tests = 0;
	for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < n_surv; t += stride) {
		int32_t s = surv[t];
		double ra = sec.ra[s], dec = sec.dec[s];
		int32_t ci, cj;
		if (cp.scheme == NWAYHIP_SCHEME_FLAT)
			flat_cell(ra, dec, cp.err_deg, ci, cj);
		else
			sphere_cell(ra, dec, cp, ci, cj);
		unsigned long long key = pack_key(ci, cj);
		uint32_t slot = hash_slot((uint32_t)ci, (uint32_t)cj) & ht.slot_mask;
		bool have_point = false;
		SkyPoint b;
		for (uint32_t probe = 0; probe <= ht.slot_mask; ++probe) {
			unsigned long long k = ht.keys[slot];
			if (k == EMPTY_KEY) break;
			if (k == key) {
				int32_t p = ht.vals[slot];
				double sep = 0.0;
				bool keep = true;
				if (radius_filter) {
					if (!have_point) {
						b = sky_point(ra, dec);
						have_point = true;
					}
					SkyPoint a;
					a.lon = plon[p];
					a.slat = pslat[p];
					a.clat = pclat[p];
					sep = separation_arcsec(a, b);
					++tests;
					if (radius_filter) keep = sep < radius_arcsec;  // max_separation < match_radius
				}
				if (keep) {
					long long pos = (long long)atomicAdd((unsigned long long*)n_pairs, 1ull);
					if (pos < cap_pairs) {
						pair_p[pos] = p;
						pair_s[pos] = s;
						pair_sep[pos] = sep;
						atomicAdd(&cnt[p], 1);
					} else {
						atomicOr((unsigned long long*)&status[NWAYHIP_ST_FLAGS], (unsigned long long)NWAYHIP_FLAG_PAIR_OVERFLOW);
					}
				}
			}
			slot = (slot + 1) & ht.slot_mask;
		}
	}
	if (tests) atomicAdd((unsigned long long*)&status[NWAYHIP_ST_TESTS], tests);
}

// ------------------------------------------------------------------------------------
// scans with a length that lives on the device (3 launches: partial, top, consumer)
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BLOCK) k_scan_partial(const int32_t* __restrict__ cnt, const long long* n_ptr,
	long long n_host, long long* partial) {
	const long long n = n_ptr ? *n_ptr : n_host;
	const long long chunk = chunk_of(n, gridDim.x);
	const long long lo = (long long)blockIdx.x * chunk;
	const long long hi = min(n, lo + chunk);
	long long acc = 0;
	for (long long i = lo + threadIdx.x; i < hi; i += BLOCK) acc += cnt[i];
	long long total;
	block_exclusive_scan(acc, total);
	if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// one block: exclusive scan of the partials in place; total -> *total_out (and optional copies)
__global__ void __launch_bounds__(1024) k_scan_top(long long* partial, int nparts, long long* total_out,
	long long* total_out2, long long cap, long long* status, unsigned long long overflow_flag) {
	__shared__ long long buf[1024];
	int t = threadIdx.x;
	long long v = t < nparts ? partial[t] : 0;
	buf[t] = v;
	__syncthreads();
	for (int o = 1; o < 1024; o <<= 1) {
		long long x = t >= o ? buf[t - o] : 0;
		__syncthreads();
		buf[t] += x;
		__syncthreads();
	}
	if (t < nparts) partial[t] = buf[t] - v;
	if (t == 1023) {
		long long total = buf[1023];
		if (total_out) *total_out = total;
		if (total_out2) *total_out2 = total;
		if (cap >= 0 && total > cap) atomicOr((unsigned long long*)&status[NWAYHIP_ST_FLAGS], overflow_flag);
	}
}

// materialise exclusive offsets out[0..n] (out[n] = total) from counts + scanned partials
__global__ void __launch_bounds__(BLOCK) k_scan_write(const int32_t* __restrict__ cnt, long long n,
	const long long* __restrict__ partial, long long* __restrict__ out) {
	const long long chunk = chunk_of(n, gridDim.x);
	const long long lo = (long long)blockIdx.x * chunk;
	const long long hi = min(n, lo + chunk);
	long long base = partial[blockIdx.x];
	for (long long i0 = lo; i0 < hi; i0 += BLOCK) {
		long long i = i0 + threadIdx.x;
		long long v = i < hi ? cnt[i] : 0;
		long long total;
		long long ex = block_exclusive_scan(v, total);
		if (i < hi) out[i] = base + ex;
		base += total;
	}
	// the block owning the tail writes the total
	const bool owner = (n == 0) ? (blockIdx.x == 0) : (lo < n && lo + chunk >= n);
	if (owner && threadIdx.x == 0) out[n] = base;
}

// ------------------------------------------------------------------------------------
// neighbour lists: scatter links into per-primary segments, then order each segment by
// secondary index (the sorted() of fastskymatch.py:181,217 restricted to one primary)
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BLOCK) k_scatter(const int32_t* __restrict__ pair_p, const int32_t* __restrict__ pair_s,
	const double* __restrict__ pair_sep, const long long* n_pairs_ptr, long long cap_pairs,
	const long long* __restrict__ off, int32_t* cnt, int32_t* tmp_s, double* tmp_sep) {
	const long long n = min(*n_pairs_ptr, cap_pairs);
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < n; e += stride) {
		int32_t p = pair_p[e];
		int slot = atomicSub(&cnt[p], 1) - 1;  // cnt counts down to 0: reusable as "fill"
		long long pos = off[p] + slot;
		tmp_s[pos] = pair_s[e];
		tmp_sep[pos] = pair_sep[e];
	}
}

// one wave per primary: rank every link by counting smaller secondary indices
__global__ void __launch_bounds__(BLOCK) k_segment_order(long long n_primary, const long long* __restrict__ off,
	const int32_t* __restrict__ tmp_s, const double* __restrict__ tmp_sep, int32_t* __restrict__ list_s,
	double* __restrict__ list_sep) {
	const int lane = lane_id();
	const long long wave0 = ((long long)blockIdx.x * BLOCK + threadIdx.x) / WAVE;
	const long long nwaves = (long long)gridDim.x * BLOCK / WAVE;
	for (long long p = wave0; p < n_primary; p += nwaves) {
		const long long lo = off[p];
		const int n = (int)(off[p + 1] - lo);
		if (n == 0) continue;
		if (n <= WAVE) {
			int32_t s = lane < n ? tmp_s[lo + lane] : 0x7fffffff;
			double sep = lane < n ? tmp_sep[lo + lane] : 0.0;
			int rank = 0;
			for (int m = 0; m < n; ++m) rank += (__shfl(s, m) < s) ? 1 : 0;
			if (lane < n) {
				list_s[lo + rank] = s;
				list_sep[lo + rank] = sep;
			}
		} else {
			for (int e = lane; e < n; e += WAVE) {
				int32_t s = tmp_s[lo + e];
				int rank = 0;
				for (int m = 0; m < n; ++m) rank += (tmp_s[lo + m] < s) ? 1 : 0;
				list_s[lo + rank] = s;
				list_sep[lo + rank] = tmp_sep[lo + e];
			}
		}
	}
}

// ------------------------------------------------------------------------------------
// tuple expansion, one catalogue per level     fastskymatch.py:174-181 (itertools.product
// of [-1] + sorted list per catalogue) restricted to combinations that share a bucket
// (cell span <= 1 over all present members) and, with the radius filter, whose pairwise
// separations are all < radius (__init__.py:166,180).  Items keep lexicographic order.
// ------------------------------------------------------------------------------------

struct ExpandArgs {
	int level;                 // number of secondary catalogues already in the item (>= 0)
	int scheme;
	int radius_filter;
	double err_deg;
	double radius_arcsec;
	const int32_t* in_idx[NWAYHIP_MAXCAT];   // [0] may be NULL at level 0 (item t == primary t)
	int32_t* out_idx[NWAYHIP_MAXCAT];
	const long long* off;      // neighbour list of the catalogue being added
	const int32_t* list_s;
	CatView newcat;
	Cats cats;
};

template <bool FILL>
__device__ __forceinline__ int expand_item(const ExpandArgs& a, long long t, long long out_pos, long long cap_rows) {
	const int32_t p = a.in_idx[0] ? a.in_idx[0][t] : (int32_t)t;
	const long long lo = a.off[p], hi = a.off[p + 1];
	int32_t member[NWAYHIP_MAXCAT];
	member[0] = p;
	int npresent = 0;
	for (int c = 1; c <= a.level; ++c) {
		member[c] = a.in_idx[c][t];
		npresent += member[c] >= 0;
	}
	int count = 1;
	if (FILL && out_pos < cap_rows) {
		for (int c = 0; c <= a.level; ++c) a.out_idx[c][out_pos] = member[c];
		a.out_idx[a.level + 1][out_pos] = -1;
	}
	if (npresent == 0) {
		if (FILL) {
			for (long long e = lo; e < hi; ++e) {
				long long pos = out_pos + 1 + (e - lo);
				if (pos < cap_rows) {
					for (int c = 0; c <= a.level; ++c) a.out_idx[c][pos] = member[c];
					a.out_idx[a.level + 1][pos] = a.list_s[e];
				}
			}
		}
		return 1 + (int)(hi - lo);
	}
	// earlier secondaries of this item
	SkyPoint pts[NWAYHIP_MAXCAT];
	int32_t ci[NWAYHIP_MAXCAT], cj[NWAYHIP_MAXCAT];
	for (int c = 1; c <= a.level; ++c) {
		if (member[c] < 0) continue;
		double ra = a.cats.c[c].ra[member[c]], dec = a.cats.c[c].dec[member[c]];
		if (a.radius_filter) pts[c] = sky_point(ra, dec);
		if (a.scheme == NWAYHIP_SCHEME_FLAT) flat_cell(ra, dec, a.err_deg, ci[c], cj[c]);
	}
	for (long long e = lo; e < hi; ++e) {
		const int32_t s = a.list_s[e];
		double ra = a.newcat.ra[s], dec = a.newcat.dec[s];
		bool ok = true;
		int32_t si = 0, sj = 0;
		if (a.scheme == NWAYHIP_SCHEME_FLAT) flat_cell(ra, dec, a.err_deg, si, sj);
		SkyPoint b;
		if (a.radius_filter) b = sky_point(ra, dec);
		for (int c = 1; c <= a.level && ok; ++c) {
			if (member[c] < 0) continue;
			if (a.scheme == NWAYHIP_SCHEME_FLAT) ok = (abs(ci[c] - si) <= 1) && (abs(cj[c] - sj) <= 1);
			if (ok && a.radius_filter) ok = separation_arcsec(pts[c], b) < a.radius_arcsec;
		}
		if (ok) {
			if (FILL) {
				long long pos = out_pos + count;
				if (pos < cap_rows) {
					for (int c = 0; c <= a.level; ++c) a.out_idx[c][pos] = member[c];
					a.out_idx[a.level + 1][pos] = s;
				}
			}
			++count;
		}
	}
	return count;
}

__global__ void __launch_bounds__(BLOCK) k_expand_count(ExpandArgs a, const long long* n_items_ptr, int32_t* cnt) {
	const long long n = *n_items_ptr;
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < n; t += stride)
		cnt[t] = expand_item<false>(a, t, 0, 0);
}

// same chunking as k_scan_partial: block b owns items [b*chunk, (b+1)*chunk)
__global__ void __launch_bounds__(BLOCK) k_expand_fill(ExpandArgs a, const long long* n_items_ptr,
	const int32_t* __restrict__ cnt, const long long* __restrict__ partial, long long cap_rows,
	long long* group_start_out, long long n_primary) {
	const long long n = *n_items_ptr;
	const long long chunk = chunk_of(n, gridDim.x);
	const long long lo = (long long)blockIdx.x * chunk;
	const long long hi = min(n, lo + chunk);
	long long base = partial[blockIdx.x];
	for (long long i0 = lo; i0 < hi; i0 += BLOCK) {
		long long t = i0 + threadIdx.x;
		long long v = t < hi ? cnt[t] : 0;
		long long total;
		long long pos = base + block_exclusive_scan(v, total);
		if (t < hi) {
			expand_item<true>(a, t, pos, cap_rows);
			// first item of a primary marks the start of its group
			int32_t p = a.in_idx[0] ? a.in_idx[0][t] : (int32_t)t;
			bool first = (t == 0) || ((a.in_idx[0] ? a.in_idx[0][t - 1] : (int32_t)(t - 1)) != p);
			if (first && group_start_out) group_start_out[p] = pos;
		}
		base += total;
	}
}

// ------------------------------------------------------------------------------------
// K4  rows: separations, Separation_max, ncat (__init__.py:143-177), log_bf
//     (bayesdistance.py:64-86 per presence pattern, __init__.py:234-252), prior (:254),
//     dist_post (:110), log_post_weight (:410, stored in p_i until the group pass)
// ------------------------------------------------------------------------------------

struct RowArgs {
	int ncat;
	Cats cats;
	int32_t* idx[NWAYHIP_MAXCAT];
	double* sep[NWAYHIP_MAXPAIR];
	double* sep_max;
	int8_t* ncat_out;
	double* log_bf;
	double* log_bf_corrected;
	double* prior;
	double* dist_post;
	double* p_single;  // may be NULL
	double* lpw;  // = p_i column, temporarily
	double prior_table[1 << (NWAYHIP_MAXCAT - 1)];
};

__global__ void __launch_bounds__(BLOCK) k_rows(RowArgs a, const long long* n_rows_ptr, long long cap_rows) {
	const long long n = min(*n_rows_ptr, cap_rows);
	const long long stride = (long long)gridDim.x * BLOCK;
	const double log_arcsec2rad = log(3600 * 180 / M_PI);  // bayesdistance.py:15
	const double log10e = log10(M_E);
	for (long long r = (long long)blockIdx.x * BLOCK + threadIdx.x; r < n; r += stride) {
		int32_t member[NWAYHIP_MAXCAT];
		SkyPoint pts[NWAYHIP_MAXCAT];
		double w[NWAYHIP_MAXCAT];
		int npresent = 0;
		unsigned pattern = 0;
		for (int c = 0; c < a.ncat; ++c) {
			int32_t m = a.idx[c][r];
			member[c] = m;
			if (m >= 0) {
				pts[c] = sky_point(a.cats.c[c].ra[m], a.cats.c[c].dec[m]);
				double s = a.cats.c[c].sigma ? a.cats.c[c].sigma[m] : a.cats.c[c].sigma_const;
				w[c] = pow(s, -2.);  // w = s**-2.
				++npresent;
				if (c > 0) pattern |= 1u << (c - 1);
			}
		}
		// pairwise separations, q = sum_{i<j} w_i w_j p_ij^2
		double sepmax = 0.0, q = 0.0;
		int pi = 0;
		for (int i = 0; i < a.ncat; ++i)
			for (int j = i + 1; j < a.ncat; ++j, ++pi) {
				double sep = nan("");
				if (member[i] >= 0 && member[j] >= 0) {
					sep = separation_arcsec(pts[i], pts[j]);
					sepmax = fmax(sepmax, sep);
					q += w[i] * w[j] * (sep * sep);
				}
				a.sep[pi][r] = sep;
			}
		double wsum = 0.0, slogw = 0.0;
		for (int c = 0; c < a.ncat; ++c)
			if (member[c] >= 0) {
				wsum += w[c];
				slogw += log(w[c]);
			}
		double norm = (npresent - 1) * log(2.0) + 2 * (npresent - 1) * log_arcsec2rad;
		double slog = slogw - log(wsum);
		double exponent = -q / 2 / wsum;
		double logbf = (norm + slog + exponent) * log10e;
		double prior = a.prior_table[pattern];
		a.sep_max[r] = sepmax;
		a.ncat_out[r] = (int8_t)npresent;
		a.log_bf[r] = logbf;
		if (a.log_bf_corrected) a.log_bf_corrected[r] = logbf;
		a.prior[r] = prior;
		double post = posterior_ref(prior, logbf);
		a.dist_post[r] = post;
		if (a.p_single) a.p_single[r] = post;  // total == dist_bayesfactor when there are no biases
		a.lpw[r] = logbf + log10(prior);
	}
}

// ------------------------------------------------------------------------------------
// K7  per-primary group statistics        __init__.py:423-457 == nway.py:547-578
//     one wave64 per primary; shuffle reductions (fixed order => deterministic)
// ------------------------------------------------------------------------------------

struct GroupArgs {
	long long n_groups;
	const long long* group_start;  // [n_groups + 1]
	const double* total;           // may be NULL: then lpw_inout already holds log_post_weight
	const double* prior;
	double ratio;
	double* p_single;              // may be NULL
	double* p_any;
	double* p_i;                   // in: log_post_weight when total == NULL
	int8_t* match_flag;
	long long cap_rows;
};

__global__ void __launch_bounds__(BLOCK) k_groups(GroupArgs a) {
	const int lane = lane_id();
	const long long wave0 = ((long long)blockIdx.x * BLOCK + threadIdx.x) / WAVE;
	const long long nwaves = (long long)gridDim.x * BLOCK / WAVE;
	const double ninf = -INFINITY;
	for (long long g = wave0; g < a.n_groups; g += nwaves) {
		const long long lo = a.group_start[g];
		const long long hi = min(a.group_start[g + 1], a.cap_rows);
		const long long n = hi - lo;
		if (n <= 0) continue;
		// pass 1: log_post_weight (and p_single), maxima over all rows / rows 1..
		double mx = ninf, mx1 = ninf;
		for (long long r = lo + lane; r < hi; r += WAVE) {
			double v;
			if (a.total) {
				double tot = a.total[r], pr = a.prior[r];
				v = tot + log10(pr);                       // unnormalised_log_posterior
				if (a.p_single) a.p_single[r] = posterior_ref(pr, tot);
				a.p_i[r] = v;
			} else {
				v = a.p_i[r];
			}
			mx = fmax(mx, v);
			if (r > lo) mx1 = fmax(mx1, v);
		}
		mx = wave_max(mx);
		mx1 = wave_max(mx1);
		// pass 2: sums of 10^(v - offset)
		double sum = 0.0, sum1 = 0.0;
		for (long long r = lo + lane; r < hi; r += WAVE) {
			double v = a.p_i[r];
			sum += exp10_ref(v - mx);
			if (r > lo) sum1 += exp10_ref(v - mx1);
		}
		sum = wave_sum(sum);
		sum1 = wave_sum(sum1);
		const double bfsum = log10(sum) + mx;
		const double bfsum1 = (n > 1) ? log10(sum1) + mx1 : 0.0;
		const double p_none = a.p_i[lo];
		const double p_any = 1 - exp10_ref(p_none - bfsum);
		// pass 3: p_i and its maximum
		double best = 0.0;  // p_i[0] = 0 is always a member
		for (long long r = lo + lane; r < hi; r += WAVE) {
			double pi = (r == lo) ? 0.0 : exp10_ref(a.p_i[r] - bfsum1);
			a.p_i[r] = pi;
			a.p_any[r] = p_any;
			best = fmax(best, pi);
		}
		best = wave_max(best);
		// pass 4: flags
		for (long long r = lo + lane; r < hi; r += WAVE) {
			double pi = a.p_i[r];
			a.match_flag[r] = (best == pi) ? 1 : ((pi > a.ratio * best) ? 2 : 0);
		}
	}
}

// ------------------------------------------------------------------------------------
// K5  unrelated-association correction with the behaviour of the script, nway.py:366-420.
//     Row i with ncat <= k-2: over rows j of the same primary with ncat[j] > 2, the catalogues
//     absent in i but present in j (>= 2 of them) form a sub-association; its
//     log_bf + log10(nu_first / prod nu+) maximised, added to row i if positive.
// ------------------------------------------------------------------------------------

struct CorrArgs {
	int ncat;
	long long n_groups;
	const long long* group_start;
	int32_t* idx[NWAYHIP_MAXCAT];
	double* sep[NWAYHIP_MAXPAIR];
	const int8_t* ncat_row;
	Cats cats;
	double dens[NWAYHIP_MAXCAT];
	double dens_plus[NWAYHIP_MAXCAT];
	const double* log_bf;
	double* log_bf_corrected;
	const double* prior;
	double* dist_post;
	double* p_single;  // may be NULL
	double* lpw;
	long long cap_rows;
};

__device__ __forceinline__ int pair_index(int i, int j, int k) {  // i < j
	return i * k - i * (i + 1) / 2 + (j - i - 1);
}

__global__ void __launch_bounds__(BLOCK) k_correct(CorrArgs a) {
	const int lane = lane_id();
	const long long wave0 = ((long long)blockIdx.x * BLOCK + threadIdx.x) / WAVE;
	const long long nwaves = (long long)gridDim.x * BLOCK / WAVE;
	const double log_arcsec2rad = log(3600 * 180 / M_PI);
	const double log10e = log10(M_E);
	const int k = a.ncat;
	for (long long g = wave0; g < a.n_groups; g += nwaves) {
		const long long lo = a.group_start[g];
		const long long hi = min(a.group_start[g + 1], a.cap_rows);
		for (long long i = lo; i < hi; ++i) {
			if (a.ncat_row[i] > k - 2) continue;  // wave-uniform
			unsigned missing = 0;
			for (int c = 1; c < k; ++c)
				if (a.idx[c][i] < 0) missing |= 1u << c;
			double best = 0.0;
			for (long long j = lo + lane; j < hi; j += WAVE) {
				if (!(a.ncat_row[j] > 2)) continue;
				int aug[NWAYHIP_MAXCAT];
				int na = 0;
				for (int c = 1; c < k; ++c)
					if (((missing >> c) & 1u) && a.idx[c][j] >= 0) aug[na++] = c;
				if (na < 2) continue;
				double w[NWAYHIP_MAXCAT];
				double wsum = 0.0, slogw = 0.0, densprod = 1.0;
				for (int x = 0; x < na; ++x) {
					int c = aug[x];
					int32_t m = a.idx[c][j];
					double s = a.cats.c[c].sigma ? a.cats.c[c].sigma[m] : a.cats.c[c].sigma_const;
					w[x] = pow(s, -2.);
					wsum += w[x];
					slogw += log(w[x]);
					densprod *= a.dens_plus[c];
				}
				double q = 0.0;
				for (int x = 0; x < na; ++x)
					for (int y = x + 1; y < na; ++y) {
						double sep = a.sep[pair_index(aug[x], aug[y], k)][j];
						q += w[x] * w[y] * (sep * sep);
					}
				double norm = (na - 1) * log(2.0) + 2 * (na - 1) * log_arcsec2rad;
				double logbf = (norm + (slogw - log(wsum)) + (-q / 2 / wsum)) * log10e;
				double prior_j = a.dens[aug[0]] / densprod;
				double logpost = logbf + log10(prior_j);
				if (logpost > best) best = logpost;
			}
			best = wave_max(best);
			if (lane == 0 && best > 0) {
				double lb = a.log_bf[i] + best;
				double pr = a.prior[i];
				double post = posterior_ref(pr, lb);
				a.log_bf_corrected[i] = lb;
				a.dist_post[i] = post;
				if (a.p_single) a.p_single[i] = post;
				a.lpw[i] = lb + log10(pr);
			}
		}
	}
}

// ------------------------------------------------------------------------------------
// elementwise kernels behind the reference's array functions
// ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(BLOCK) k_dist(const double* a_ra, const double* a_dec, const double* b_ra,
	const double* b_dec, long long n, double* out) {
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
		SkyPoint a = sky_point(a_ra[i], a_dec[i]), b = sky_point(b_ra[i], b_dec[i]);
		double dlon = b.lon - a.lon;
		double sdlon, cdlon;
		sincos(dlon, &sdlon, &cdlon);
		double num1 = b.clat * sdlon;
		double num2 = a.clat * b.slat - a.slat * b.clat * cdlon;
		double den = a.slat * b.slat + a.clat * b.clat * cdlon;
		out[i] = atan2(hypot(num1, num2), den) * 180 / M_PI;
	}
}

struct LogBfArgs {
	int ncat;
	const double* sep[NWAYHIP_MAXCAT * NWAYHIP_MAXCAT];
	const double* sigma[NWAYHIP_MAXCAT];
};

__global__ void __launch_bounds__(BLOCK) k_log_bf(LogBfArgs a, long long n, double* out) {
	const long long stride = (long long)gridDim.x * BLOCK;
	const double log_arcsec2rad = log(3600 * 180 / M_PI);
	const double log10e = log10(M_E);
	for (long long r = (long long)blockIdx.x * BLOCK + threadIdx.x; r < n; r += stride) {
		double w[NWAYHIP_MAXCAT];
		double wsum = 0.0, slogw = 0.0, q = 0.0;
		for (int c = 0; c < a.ncat; ++c) {
			w[c] = pow(a.sigma[c][r], -2.);
			wsum += w[c];
			slogw += log(w[c]);
		}
		for (int i = 0; i < a.ncat; ++i)
			for (int j = i + 1; j < a.ncat; ++j) {
				double p = a.sep[i * a.ncat + j][r];
				q += w[i] * w[j] * (p * p);
			}
		double norm = (a.ncat - 1) * log(2.0) + 2 * (a.ncat - 1) * log_arcsec2rad;
		out[r] = (norm + (slogw - log(wsum)) + (-q / 2 / wsum)) * log10e;
	}
}

__global__ void __launch_bounds__(BLOCK) k_posterior(int mode, const double* prior, const double* logbf, long long n,
	double* out) {
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
		double pr = prior[i], lb = logbf[i];
		double v;
		if (mode == 0)
			v = posterior_ref(pr, lb);
		else if (mode == 1)
			v = -log10(1 + (1 - pr) * exp10_ref(-lb - log10(pr)));
		else
			v = lb + log10(pr);
		out[i] = v;
	}
}

__global__ void __launch_bounds__(BLOCK) k_extent(const double* ra, const double* dec, long long n, double* out4) {
	// out4 pre-initialised by the host wrapper: {+inf, -inf, 0, 0}; doubles >= 0 compare like
	// their bit patterns, RA may be negative, so use CAS loops
	const long long stride = (long long)gridDim.x * BLOCK;
	double lo = INFINITY, hi = -INFINITY, ad = 0.0;
	long long nnan = 0;
	for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
		double r = ra[i], d = dec[i];
		if (!(r == r) || !(d == d)) {
			++nnan;
			continue;
		}
		lo = fmin(lo, r);
		hi = fmax(hi, r);
		ad = fmax(ad, fabs(d));
	}
	lo = -wave_max(-lo);
	hi = wave_max(hi);
	ad = wave_max(ad);
	double nn = wave_sum((double)nnan);
	if (lane_id() == 0) {
		unsigned long long* p = (unsigned long long*)out4;
		unsigned long long old = p[0];
		while (__longlong_as_double(old) > lo) {
			unsigned long long prev = atomicCAS(&p[0], old, (unsigned long long)__double_as_longlong(lo));
			if (prev == old) break;
			old = prev;
		}
		old = p[1];
		while (__longlong_as_double(old) < hi) {
			unsigned long long prev = atomicCAS(&p[1], old, (unsigned long long)__double_as_longlong(hi));
			if (prev == old) break;
			old = prev;
		}
		old = p[2];
		while (__longlong_as_double(old) < ad) {
			unsigned long long prev = atomicCAS(&p[2], old, (unsigned long long)__double_as_longlong(ad));
			if (prev == old) break;
			old = prev;
		}
		if (nn > 0) atomicAdd(&out4[3], nn);
	}
}

__global__ void k_set_counter(long long* dst, long long value) { *dst = value; }

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

int grid_for(long long n, int per_thread = 1, int max_blocks = 256 * 8) {
	long long b = (n + (long long)BLOCK * per_thread - 1) / ((long long)BLOCK * per_thread);
	if (b < 1) b = 1;
	if (b > max_blocks) b = max_blocks;
	return (int)b;
}

uint64_t next_pow2(uint64_t x) {
	uint64_t p = 1;
	while (p < x) p <<= 1;
	return p;
}

}  // namespace

struct nwayhip_plan {
	nwayhip_match_params prm;
	int64_t n[NWAYHIP_MAXCAT];
	int64_t cap_pairs, cap_rows;
	CellParams cp;
	uint64_t nslots, nbits;
	// workspace layout (byte offsets)
	size_t o_keys, o_vals, o_zero_begin, o_bitmap, o_cnt[NWAYHIP_MAXCAT], o_zero_end;
	size_t o_plon, o_pslat, o_pclat;
	size_t o_surv, o_pair_p, o_pair_s, o_pair_sep, o_tmp_s, o_tmp_sep;
	size_t o_off[NWAYHIP_MAXCAT], o_list_s[NWAYHIP_MAXCAT], o_list_sep[NWAYHIP_MAXCAT];
	size_t o_partial, o_item_cnt, o_counters, o_idx[2];
	size_t total_bytes;
};

extern "C" {

int nwayhip_version(void) { return NWAYHIP_ABI_VERSION; }

const char* nwayhip_last_error(void) { return g_err; }

int nwayhip_device_count(int* h_count) {
	if (!h_count) return fail("nwayhip_device_count: null argument");
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) {
		*h_count = 0;
		return fail("hipGetDeviceCount: %s", hipGetErrorString(e));
	}
	*h_count = n;
	return 0;
}

int nwayhip_dist(const double* a_ra, const double* a_dec, const double* b_ra, const double* b_dec, int64_t n,
	double* out_deg, void* stream) {
	if (n < 0) return fail("nwayhip_dist: negative length");
	if (n == 0) return 0;
	hipLaunchKernelGGL(k_dist, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, a_ra, a_dec, b_ra, b_dec,
		(long long)n, out_deg);
	HIP_TRY(hipGetLastError());
	return 0;
}

int nwayhip_log_bf(int32_t ncat, int64_t n, const double* const* h_sep, const double* const* h_sigma, double* out,
	void* stream) {
	if (ncat < 1 || ncat > NWAYHIP_MAXCAT) return fail("nwayhip_log_bf: ncat must be 1..%d", NWAYHIP_MAXCAT);
	if (n < 0) return fail("nwayhip_log_bf: negative length");
	if (n == 0) return 0;
	LogBfArgs a;
	memset(&a, 0, sizeof(a));
	a.ncat = ncat;
	for (int i = 0; i < ncat; ++i) {
		a.sigma[i] = h_sigma[i];
		for (int j = i + 1; j < ncat; ++j) {
			a.sep[i * ncat + j] = h_sep[i * ncat + j];
			if (!a.sep[i * ncat + j]) return fail("nwayhip_log_bf: separation (%d,%d) is null", i, j);
		}
	}
	hipLaunchKernelGGL(k_log_bf, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, a, (long long)n, out);
	HIP_TRY(hipGetLastError());
	return 0;
}

int nwayhip_posterior(int32_t mode, const double* prior, const double* log_bf, int64_t n, double* out, void* stream) {
	if (mode < 0 || mode > 2) return fail("nwayhip_posterior: mode must be 0, 1 or 2");
	if (n <= 0) return n < 0 ? fail("nwayhip_posterior: negative length") : 0;
	hipLaunchKernelGGL(k_posterior, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, mode, prior, log_bf,
		(long long)n, out);
	HIP_TRY(hipGetLastError());
	return 0;
}

int nwayhip_catalogue_extent(const double* ra, const double* dec, int64_t n, double* d_out4, void* stream) {
	const double init[4] = {INFINITY, -INFINITY, 0.0, 0.0};
	HIP_TRY(hipMemcpyAsync(d_out4, init, sizeof(init), hipMemcpyHostToDevice, (hipStream_t)stream));
	if (n > 0) {
		hipLaunchKernelGGL(k_extent, dim3(grid_for(n, 4)), dim3(BLOCK), 0, (hipStream_t)stream, ra, dec, (long long)n,
			d_out4);
		HIP_TRY(hipGetLastError());
	}
	return 0;
}

int nwayhip_group_stats(int64_t n_rows, int64_t n_groups, const int64_t* group_start, const double* total,
	const double* prior, double prob_ratio_secondary, double* p_single, double* p_any, double* p_i, int8_t* match_flag,
	void* stream) {
	if (n_rows < 0 || n_groups < 0) return fail("nwayhip_group_stats: negative size");
	if (n_groups == 0) return 0;
	if (!total || !prior || !group_start || !p_any || !p_i || !match_flag)
		return fail("nwayhip_group_stats: null argument");
	GroupArgs g;
	g.n_groups = n_groups;
	g.group_start = (const long long*)group_start;
	g.total = total;
	g.prior = prior;
	g.ratio = prob_ratio_secondary;
	g.p_single = p_single;
	g.p_any = p_any;
	g.p_i = p_i;
	g.match_flag = match_flag;
	g.cap_rows = n_rows;
	hipLaunchKernelGGL(k_groups, dim3(grid_for(n_groups * WAVE)), dim3(BLOCK), 0, (hipStream_t)stream, g);
	HIP_TRY(hipGetLastError());
	return 0;
}

int nwayhip_plan_create(nwayhip_plan** out, const nwayhip_match_params* prm, const int64_t* h_n, int64_t cap_pairs,
	int64_t cap_rows) {
	if (!out || !prm || !h_n) return fail("nwayhip_plan_create: null argument");
	if (prm->ncat < 2 || prm->ncat > NWAYHIP_MAXCAT) return fail("ncat must be 2..%d", NWAYHIP_MAXCAT);
	if (prm->scheme != NWAYHIP_SCHEME_FLAT && prm->scheme != NWAYHIP_SCHEME_SPHERE) return fail("unknown scheme %d", prm->scheme);
	if (prm->scheme == NWAYHIP_SCHEME_SPHERE && !prm->radius_filter)
		return fail("the all-sky scheme is only defined with the radius filter");
	if (!(prm->err_deg > 0) || !(prm->radius_arcsec > 0)) return fail("radius must be positive");
	if (cap_pairs < 1 || cap_rows < 1) return fail("capacities must be positive");
	if (cap_rows >= (1ll << 31) || cap_pairs >= (1ll << 31)) return fail("capacities must be < 2^31");
	for (int c = 0; c < prm->ncat; ++c)
		if (h_n[c] < 0 || h_n[c] >= (1ll << 31)) return fail("catalogue %d: size out of range", c);
	if (h_n[0] < 1) return fail("primary catalogue is empty");
	if (prm->scheme == NWAYHIP_SCHEME_FLAT && 360.0 / prm->err_deg >= 2147483000.0)
		return fail("radius too small for 32-bit flat cells");
	nwayhip_plan* pl = new (std::nothrow) nwayhip_plan;
	if (!pl) return fail("out of host memory");
	memset(pl, 0, sizeof(*pl));
	pl->prm = *prm;
	for (int c = 0; c < prm->ncat; ++c) pl->n[c] = h_n[c];
	pl->cap_pairs = cap_pairs;
	pl->cap_rows = cap_rows;
	const int k = prm->ncat;
	const int64_t n0 = h_n[0];

	CellParams& cp = pl->cp;
	cp.scheme = prm->scheme;
	cp.err_deg = prm->err_deg;
	cp.ra_lo = 10 * prm->err_deg;
	cp.ra_hi = 360 - 10 * prm->err_deg;
	double radius_deg = prm->radius_arcsec / 3600.0;
	cp.reach_deg = radius_deg * (1 + 1e-9) + 1e-12;
	double factor = prm->sphere_cell_factor > 0 ? prm->sphere_cell_factor : 8.0;
	if (factor < 2.0) factor = 2.0;
	double cell = radius_deg * factor;
	if (cell > 30.0) cell = 30.0;
	if (cell < 360.0 / 2.0e9) cell = 360.0 / 2.0e9;
	cp.nbands = (int)floor(180.0 / cell);
	if (cp.nbands < 1) cp.nbands = 1;
	cp.inv_h = cp.nbands / 180.0;
	cp.nra_eq = (int)floor(360.0 / cell);
	if (cp.nra_eq < 1) cp.nra_eq = 1;
	double h = 180.0 / cp.nbands;
	cp.taper = (int)ceil((180.0 / M_PI) / h);  // bands within 1 rad of a pole
	if (cp.taper < 1) cp.taper = 1;

	uint64_t regs = (prm->scheme == NWAYHIP_SCHEME_FLAT) ? 9ull * n0 : 12ull * n0;
	pl->nslots = next_pow2(2 * regs + 64);
	if (pl->nslots > (1ull << 32)) {
		delete pl;
		return fail("primary catalogue too large for the cell table");
	}
	uint64_t nbits = prm->bitmap_bits > 0 ? next_pow2((uint64_t)prm->bitmap_bits) : (1ull << 24);
	if (prm->bitmap_bits <= 0)
		while (nbits < 32 * regs && nbits < (1ull << 31)) nbits <<= 1;
	if (nbits < 1024) nbits = 1024;
	if (nbits > (1ull << 32)) nbits = 1ull << 32;
	pl->nbits = nbits;

	size_t o = 0;
	auto take = [&](size_t bytes) {
		size_t at = o;
		o = align_up(o + bytes);
		return at;
	};
	int64_t nmax = 0;
	for (int c = 1; c < k; ++c) nmax = h_n[c] > nmax ? h_n[c] : nmax;
	pl->o_keys = take(pl->nslots * 8);
	pl->o_vals = take(pl->nslots * 4);
	pl->o_zero_begin = o;
	pl->o_bitmap = take(nbits / 8);
	for (int c = 1; c < k; ++c) pl->o_cnt[c] = take((n0 + 1) * 4);
	pl->o_counters = take(64 * 8);
	pl->o_zero_end = o;
	pl->o_plon = take(n0 * 8);
	pl->o_pslat = take(n0 * 8);
	pl->o_pclat = take(n0 * 8);
	pl->o_surv = take((size_t)(nmax + 2) * 4);
	pl->o_pair_p = take((size_t)cap_pairs * 4);
	pl->o_pair_s = take((size_t)cap_pairs * 4);
	pl->o_pair_sep = take((size_t)cap_pairs * 8);
	pl->o_tmp_s = take((size_t)cap_pairs * 4);
	pl->o_tmp_sep = take((size_t)cap_pairs * 8);
	for (int c = 1; c < k; ++c) {
		pl->o_off[c] = take((n0 + 1) * 8);
		pl->o_list_s[c] = take((size_t)cap_pairs * 4);
		pl->o_list_sep[c] = take((size_t)cap_pairs * 8);
	}
	pl->o_partial = take(1024 * 8);
	pl->o_item_cnt = take((size_t)cap_rows * 4);
	if (k >= 3) pl->o_idx[0] = take((size_t)cap_rows * 4 * (k - 1));
	if (k >= 4) pl->o_idx[1] = take((size_t)cap_rows * 4 * (k - 1));
	pl->total_bytes = o;
	*out = pl;
	return 0;
}

int nwayhip_plan_destroy(nwayhip_plan* plan) {
	delete plan;
	return 0;
}

size_t nwayhip_plan_workspace_bytes(const nwayhip_plan* plan) { return plan ? plan->total_bytes : 0; }

int nwayhip_match_enqueue(nwayhip_plan* pl, const nwayhip_catalogue* h_cats, void* workspace, size_t workspace_bytes,
	const nwayhip_table* tab, int64_t* d_status, void* stream_) {
	if (!pl || !h_cats || !workspace || !tab || !d_status) return fail("nwayhip_match_enqueue: null argument");
	if (workspace_bytes < pl->total_bytes)
		return fail("workspace too small: %zu bytes given, %zu needed", workspace_bytes, pl->total_bytes);
	if (((uintptr_t)workspace & 255) != 0) return fail("workspace must be 256-byte aligned");
	if (tab->capacity < pl->cap_rows) return fail("table capacity %lld < planned %lld", (long long)tab->capacity, (long long)pl->cap_rows);
	const nwayhip_match_params& prm = pl->prm;
	const int k = prm.ncat;
	const int64_t n0 = pl->n[0];
	hipStream_t stream = (hipStream_t)stream_;
	char* ws = (char*)workspace;
	long long* status = (long long*)d_status;

	Cats cats;
	memset(&cats, 0, sizeof(cats));
	for (int c = 0; c < k; ++c) {
		if (h_cats[c].n != pl->n[c]) return fail("catalogue %d has %lld rows, plan was made for %lld", c, (long long)h_cats[c].n, (long long)pl->n[c]);
		if (pl->n[c] > 0 && (!h_cats[c].ra || !h_cats[c].dec)) return fail("catalogue %d: null coordinates", c);
		if (pl->n[c] > 0 && !h_cats[c].sigma && !(h_cats[c].sigma_const > 0)) return fail("catalogue %d: no positional error", c);
		if (((uintptr_t)h_cats[c].ra & 15) || ((uintptr_t)h_cats[c].dec & 15)) return fail("catalogue %d: coordinate columns must be 16-byte aligned", c);
		cats.c[c].ra = h_cats[c].ra;
		cats.c[c].dec = h_cats[c].dec;
		cats.c[c].sigma = h_cats[c].sigma;
		cats.c[c].sigma_const = h_cats[c].sigma_const;
		cats.c[c].n = h_cats[c].n;
	}
	for (int c = 0; c < k; ++c)
		if (!tab->idx[c]) return fail("table: idx[%d] is null", c);
	for (int p = 0; p < k * (k - 1) / 2; ++p)
		if (!tab->sep[p]) return fail("table: sep[%d] is null", p);
	if (!tab->sep_max || !tab->ncat || !tab->log_bf || !tab->prior || !tab->dist_post || !tab->p_i || !tab->group_start)
		return fail("table: null column");
	if (prm.finalize && (!tab->p_any || !tab->match_flag || !tab->p_single)) return fail("table: finalize needs p_single, p_any, match_flag");
	if (prm.correction == NWAYHIP_CORRECTION_CLI && !tab->log_bf_corrected) return fail("table: correction needs log_bf_corrected");

	HashTable ht;
	ht.keys = (unsigned long long*)(ws + pl->o_keys);
	ht.vals = (int32_t*)(ws + pl->o_vals);
	ht.slot_mask = (uint32_t)(pl->nslots - 1);
	ht.bitmap = (uint32_t*)(ws + pl->o_bitmap);
	ht.bit_mask = (uint32_t)(pl->nbits - 1);
	double* plon = (double*)(ws + pl->o_plon);
	double* pslat = (double*)(ws + pl->o_pslat);
	double* pclat = (double*)(ws + pl->o_pclat);
	long long* counters = (long long*)(ws + pl->o_counters);  // [0],[1]: item counts (ping-pong)
	long long* partial = (long long*)(ws + pl->o_partial);
	int32_t* item_cnt = (int32_t*)(ws + pl->o_item_cnt);

	HIP_TRY(hipMemsetAsync(d_status, 0, NWAYHIP_STATUS_WORDS * 8, stream));
	HIP_TRY(hipMemsetAsync(ht.keys, 0xFF, pl->nslots * 8, stream));
	HIP_TRY(hipMemsetAsync(ws + pl->o_zero_begin, 0, pl->o_zero_end - pl->o_zero_begin, stream));

	hipLaunchKernelGGL(k_register, dim3((unsigned)((n0 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, cats.c[0], pl->cp, ht, plon,
		pslat, pclat, status);

	// ---- neighbour lists per secondary catalogue
	for (int c = 1; c < k; ++c) {
		const int64_t nc = pl->n[c];
		int32_t* surv = (int32_t*)(ws + pl->o_surv);
		int32_t* cnt = (int32_t*)(ws + pl->o_cnt[c]);
		long long* off = (long long*)(ws + pl->o_off[c]);
		long long* n_surv = &status[NWAYHIP_ST_SURVIVORS + c - 1];
		long long* n_pairs = &status[NWAYHIP_ST_PAIRS + c - 1];
		if (nc > 0) {
			int grid = grid_for(nc, 8);
			if (prm.scheme == NWAYHIP_SCHEME_FLAT)
				hipLaunchKernelGGL(k_sweep<NWAYHIP_SCHEME_FLAT>, dim3(grid), dim3(BLOCK), 0, stream, cats.c[c].ra, cats.c[c].dec,
					(long long)nc, pl->cp, ht, surv, n_surv, &status[NWAYHIP_ST_NOTFLAT + c]);
			else
				hipLaunchKernelGGL(k_sweep<NWAYHIP_SCHEME_SPHERE>, dim3(grid), dim3(BLOCK), 0, stream, cats.c[c].ra, cats.c[c].dec,
					(long long)nc, pl->cp, ht, surv, n_surv, &status[NWAYHIP_ST_NOTFLAT + c]);
			hipLaunchKernelGGL(k_pairs, dim3(grid_for(nc / 16 + 1)), dim3(BLOCK), 0, stream, cats.c[c], pl->cp, ht, plon, pslat, pclat,
				surv, n_surv, prm.radius_filter, prm.radius_arcsec, (int32_t*)(ws + pl->o_pair_p), (int32_t*)(ws + pl->o_pair_s),
				(double*)(ws + pl->o_pair_sep), (long long)pl->cap_pairs, cnt, n_pairs, status);
		}
		hipLaunchKernelGGL(k_scan_partial, dim3(SCAN_BLOCKS), dim3(BLOCK), 0, stream, cnt, (const long long*)nullptr, (long long)n0, partial);
		hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, stream, partial, SCAN_BLOCKS, (long long*)nullptr, (long long*)nullptr,
			(long long)-1, status, 0ull);
		hipLaunchKernelGGL(k_scan_write, dim3(SCAN_BLOCKS), dim3(BLOCK), 0, stream, cnt, (long long)n0, partial, off);
		if (nc > 0) {
			hipLaunchKernelGGL(k_scatter, dim3(grid_for(pl->cap_pairs < nc ? pl->cap_pairs : nc)), dim3(BLOCK), 0, stream,
				(const int32_t*)(ws + pl->o_pair_p), (const int32_t*)(ws + pl->o_pair_s), (const double*)(ws + pl->o_pair_sep),
				n_pairs, (long long)pl->cap_pairs, off, cnt, (int32_t*)(ws + pl->o_tmp_s), (double*)(ws + pl->o_tmp_sep));
			hipLaunchKernelGGL(k_segment_order, dim3(grid_for(n0 * WAVE)), dim3(BLOCK), 0, stream, (long long)n0, off,
				(const int32_t*)(ws + pl->o_tmp_s), (const double*)(ws + pl->o_tmp_sep), (int32_t*)(ws + pl->o_list_s[c]),
				(double*)(ws + pl->o_list_sep[c]));
		}
	}

	// ---- breadth-first expansion
	hipLaunchKernelGGL(k_set_counter, dim3(1), dim3(1), 0, stream, &counters[0], (long long)n0);
	for (int level = 0; level < k - 1; ++level) {
		const int c = level + 1;  // catalogue being added
		const bool last = (c == k - 1);
		ExpandArgs a;
		memset(&a, 0, sizeof(a));
		a.level = level;
		a.scheme = prm.scheme;
		a.radius_filter = prm.radius_filter;
		a.err_deg = prm.err_deg;
		a.radius_arcsec = prm.radius_arcsec;
		a.off = (const long long*)(ws + pl->o_off[c]);
		a.list_s = (const int32_t*)(ws + pl->o_list_s[c]);
		a.newcat = cats.c[c];
		a.cats = cats;
		// input: level 0 -> implicit primaries; else the ping-pong buffer written by the previous level
		if (level > 0) {
			int32_t* in = (int32_t*)(ws + pl->o_idx[(level - 1) & 1]);
			for (int x = 0; x <= level; ++x) a.in_idx[x] = in + (size_t)x * pl->cap_rows;
		}
		if (last) {
			for (int x = 0; x <= c; ++x) a.out_idx[x] = tab->idx[x];
		} else {
			int32_t* outb = (int32_t*)(ws + pl->o_idx[level & 1]);
			for (int x = 0; x <= c; ++x) a.out_idx[x] = outb + (size_t)x * pl->cap_rows;
		}
		const long long* n_items = &counters[level & 1];
		long long* n_next = &counters[(level + 1) & 1];
		long long* gs_out = last ? (long long*)tab->group_start : (long long*)nullptr;
		hipLaunchKernelGGL(k_expand_count, dim3(grid_for(level == 0 ? n0 : pl->cap_rows)), dim3(BLOCK), 0, stream, a, n_items, item_cnt);
		hipLaunchKernelGGL(k_scan_partial, dim3(SCAN_BLOCKS), dim3(BLOCK), 0, stream, (const int32_t*)item_cnt, n_items, 0ll, partial);
		hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, stream, partial, SCAN_BLOCKS, n_next,
			last ? &status[NWAYHIP_ST_ROWS] : (long long*)nullptr, (long long)pl->cap_rows, status,
			(unsigned long long)NWAYHIP_FLAG_ROW_OVERFLOW);
		hipLaunchKernelGGL(k_expand_fill, dim3(SCAN_BLOCKS), dim3(BLOCK), 0, stream, a, n_items, (const int32_t*)item_cnt,
			(const long long*)partial, (long long)pl->cap_rows, gs_out, (long long)n0);
		if (last) {
			// group_start[n0] = M
			HIP_TRY(hipMemcpyAsync(&((long long*)tab->group_start)[n0], n_next, 8, hipMemcpyDeviceToDevice, stream));
		}
	}
	const long long* n_rows = &status[NWAYHIP_ST_ROWS];

	// ---- rows
	RowArgs ra;
	memset(&ra, 0, sizeof(ra));
	ra.ncat = k;
	ra.cats = cats;
	for (int c = 0; c < k; ++c) ra.idx[c] = tab->idx[c];
	for (int p = 0; p < k * (k - 1) / 2; ++p) ra.sep[p] = tab->sep[p];
	ra.sep_max = tab->sep_max;
	ra.ncat_out = tab->ncat;
	ra.log_bf = tab->log_bf;
	ra.log_bf_corrected = tab->log_bf_corrected;
	ra.prior = tab->prior;
	ra.dist_post = tab->dist_post;
	ra.p_single = prm.finalize ? tab->p_single : nullptr;
	ra.lpw = tab->p_i;
	memcpy(ra.prior_table, prm.prior_table, sizeof(ra.prior_table));
	hipLaunchKernelGGL(k_rows, dim3(grid_for(pl->cap_rows)), dim3(BLOCK), 0, stream, ra, n_rows, (long long)pl->cap_rows);

	if (prm.correction == NWAYHIP_CORRECTION_CLI && k >= 3) {
		CorrArgs ca;
		memset(&ca, 0, sizeof(ca));
		ca.ncat = k;
		ca.n_groups = n0;
		ca.group_start = (const long long*)tab->group_start;
		for (int c = 0; c < k; ++c) ca.idx[c] = tab->idx[c];
		for (int p = 0; p < k * (k - 1) / 2; ++p) ca.sep[p] = tab->sep[p];
		ca.ncat_row = tab->ncat;
		ca.cats = cats;
		memcpy(ca.dens, prm.dens, sizeof(ca.dens));
		memcpy(ca.dens_plus, prm.dens_plus, sizeof(ca.dens_plus));
		ca.log_bf = tab->log_bf;
		ca.log_bf_corrected = tab->log_bf_corrected;
		ca.prior = tab->prior;
		ca.dist_post = tab->dist_post;
		ca.p_single = prm.finalize ? tab->p_single : nullptr;
		ca.lpw = tab->p_i;
		ca.cap_rows = pl->cap_rows;
		hipLaunchKernelGGL(k_correct, dim3(grid_for(n0 * WAVE)), dim3(BLOCK), 0, stream, ca);
	}

	if (prm.finalize) {
		// total == dist_bayesfactor: p_single == dist_post (bayesdistance.posterior of the same arguments)
		GroupArgs g;
		g.n_groups = n0;
		g.group_start = (const long long*)tab->group_start;
		g.total = nullptr;
		g.prior = tab->prior;
		g.ratio = prm.prob_ratio_secondary;
		g.p_single = nullptr;
		g.p_any = tab->p_any;
		g.p_i = tab->p_i;
		g.match_flag = tab->match_flag;
		g.cap_rows = pl->cap_rows;
		hipLaunchKernelGGL(k_groups, dim3(grid_for(n0 * WAVE)), dim3(BLOCK), 0, stream, g);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

}  // extern "C"
