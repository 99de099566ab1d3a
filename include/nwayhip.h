/* nwayhip.h -- C ABI of libnwayhip.so, the MI355X (gfx950) implementation of nway's
 * match-probability hot path.
 *
 * The reference (JohannesBuchner/nway v4.7.1) is pure Python and has no FFI; the
 * boundary is its Python surface.  Every entry point below names the reference
 * interface (file:line in /root/reference) whose work it replaces; the Python host
 * (nway_amd/) keeps the reference's function names and calls these through ctypes.
 *
 * Conventions
 *   - every function returns int: 0 = ok, < 0 = error (text via nwayhip_last_error()).
 *   - no C++ types, exceptions, torch types or Python objects cross this boundary.
 *   - all array arguments are DEVICE pointers unless the name starts with h_.
 *     The caller owns every buffer (the Python host passes torch.Tensor.data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls
 *     only enqueue work; nothing synchronises with the host.
 *   - floating columns are double, SoA; catalogue row indices are int32 (N < 2^31).
 */
#ifndef NWAYHIP_H
#define NWAYHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NWAYHIP_ABI_VERSION 3            /* nwayhip_version(); 3 (round 6): nwayhip_zones_*, nwayhip_plan_profile_samples (round 5, unbumped then);
                                            2 (round 4): nwayhip_log_bf_elliptical gained f32_offsets in round 3 without a
                                            bump, NWAYHIP_ENABLE_FUSED_FRONT / NWAYHIP_FLAG_BARRIER / NWAYHIP_DESC_FUSED_FRONT are gone */
#define NWAYHIP_MAXCAT 8                 /* catalogues per match (primary + 7) */
#define NWAYHIP_MAXPAIR 28               /* MAXCAT*(MAXCAT-1)/2 separation columns */

/* candidate-cell scheme (fastskymatch.py:94-98) */
#define NWAYHIP_SCHEME_FLAT 0            /* reference's flat RA/Dec cells, :123-133 */
#define NWAYHIP_SCHEME_SPHERE 1          /* replaces the HEALPix branch :135-160 by its
                                            post-filter result (all pairwise sep < radius) */

/* unrelated-association correction */
#define NWAYHIP_CORRECTION_NONE 0        /* == nwaylib.nway_match (__init__.py:262-301 is a no-op) */
#define NWAYHIP_CORRECTION_CLI 1         /* nway.py:366-423 */

/* status words written by nwayhip_match_enqueue (device int64[NWAYHIP_STATUS_WORDS]) */
#define NWAYHIP_STATUS_WORDS 32
#define NWAYHIP_ST_ROWS 0                /* M: rows of the match table */
#define NWAYHIP_ST_FLAGS 1               /* bit mask, NWAYHIP_FLAG_* */
#define NWAYHIP_ST_REGISTRATIONS 2       /* primary -> cell registrations */
#define NWAYHIP_ST_TESTS 3               /* great-circle distance tests executed (M0') */
#define NWAYHIP_ST_REGION_NEED 4         /* with NWAYHIP_FLAG_PAIR_OVERFLOW: links one workgroup wanted to keep, if that
                                            (and not the total) is what did not fit: come back with link_region_min */
#define NWAYHIP_ST_SLOT_NEED 5           /* with NWAYHIP_FLAG_SLOT_OVERFLOW: the most candidates of one primary and catalogue (0: the
                                            overflow was not of that kind); link_slots = that many, if <= 128, keeps the sparse front */
#define NWAYHIP_ST_SURVIVORS 8           /* + c: secondaries of catalogue c passing the cell filter */
#define NWAYHIP_ST_PAIRS 16              /* + c: (primary, secondary) links of catalogue c */
#define NWAYHIP_ST_NOTFLAT 24            /* + c: 1 if catalogue c violates the flat-cell condition */

#define NWAYHIP_FLAG_PAIR_OVERFLOW 1     /* cap_pairs too small: results invalid, retry larger */
#define NWAYHIP_FLAG_ROW_OVERFLOW 2      /* cap_rows too small: results invalid, retry larger */
#define NWAYHIP_FLAG_REG_OVERFLOW 4      /* registration table too small */
#define NWAYHIP_FLAG_SLOT_OVERFLOW 8     /* a primary has more links than link_slots: repeat with more (NWAYHIP_ST_SLOT_NEED) or link_slots = -1 */
#define NWAYHIP_FLAG_LOOKBACK 16         /* the single-pass scan timed out: repeat with link_slots = -1 */
/* (32: unused since round 4 -- it was the grid barrier of round 3's registration-inside-the-sweep variant) */
#define NWAYHIP_FLAG_QUAD_DEEP 64        /* k = 3 tail with four lanes per primary (k_tail3q): a primary has three or more candidates in a
                                            catalogue: repeat with NWAYHIP_DISABLE_QUAD3 */

typedef struct nwayhip_catalogue {
	const double* ra;                    /* degrees */
	const double* dec;                   /* degrees */
	const double* sigma;                 /* positional error, arcsec (may be NULL if sigma_const > 0) */
	double sigma_const;                  /* used when sigma == NULL */
	int64_t n;
} nwayhip_catalogue;

typedef struct nwayhip_match_params {
	int32_t ncat;                        /* 2..NWAYHIP_MAXCAT */
	int32_t scheme;                      /* NWAYHIP_SCHEME_* (host decides as fastskymatch.py:94-98) */
	int32_t radius_filter;               /* 1: keep rows with Separation_max < radius (__init__.py:180);
	                                        0: raw crossproduct (fastskymatch.py:92-218; FLAT only) */
	int32_t correction;                  /* NWAYHIP_CORRECTION_* */
	int32_t finalize;                    /* 1: also run the per-primary group statistics with
	                                        total = dist_bayesfactor (no magnitude biases) */
	int32_t link_slots;                  /* sparse front (any number of catalogues; needs radius_filter): the
	                                        links of every secondary catalogue are kept in this many fixed
	                                        slots per primary (at most 128), found by the sweep itself.
	                                        0 = decide from the densities: 8 slots where every catalogue
	                                        expects < 0.5 chance neighbours lambda per primary, else the
	                                        Poisson quantile that one primary in a thousand RUNS exceeds
	                                        (+ 2), if that is at most 128 and the
	                                        primaries' cells fit the direct-mapped table; -1 = never (the
	                                        general path); > 0 = force where possible.  What follows the
	                                        front: nwayhip_plan_path() */
	double err_deg;                      /* cell size: match_radius / 60. / 60 (__init__.py:128) */
	double radius_arcsec;                /* match_radius */
	double prob_ratio_secondary;         /* __init__.py:33 */
	double dens[NWAYHIP_MAXCAT];         /* nu_c      (__init__.py:199-217) */
	double dens_plus[NWAYHIP_MAXCAT];    /* nu+_c */
	/* prior for every presence pattern: bit (c-1) set <=> catalogue c present;
	 * = nu_0 * prod(completeness[present]) / prod(nu+[present])  (__init__.py:254) */
	double prior_table[1 << (NWAYHIP_MAXCAT - 1)];
	double sphere_cell_factor;           /* all-sky cell edge in units of the radius (0 = default) */
	int64_t bitmap_bits;                 /* 0 = default; power of two */
	int64_t table_slots;                 /* cell-table slots (rounded up to a power of two); 0 = default sizing.
	                                      * NWAYHIP_FLAG_REG_OVERFLOW asks the caller to come back with more. */
	int64_t link_region_min;             /* general path: least number of links every workgroup of the pairs kernels can
	                                      * keep (0 = derived from cap_pairs); see NWAYHIP_ST_REGION_NEED */
	int64_t f32_roundtrip;               /* 1 = numerics of the script nway.py: separations pass through float32
	                                      * (FITS 'E' column, fastskymatch.py:328) before being squared in log_bf */
	/* Tuning: what tests and benchmarks use to force a path on inputs that would not take it.  All 0 =
	 * the library decides (every caller but those).  Nothing here changes a result, only which kernels
	 * produce it.  (The library reads NO environment variable unless NWAYHIP_DEV=1 is set: tools/dev.) */
	int32_t direct_log2;                 /* positions of the sparse front's direct-mapped table, log2 (10..24; > 20: the
	                                        large-table sweep) */
	int32_t fold_log2;                   /* large tables: bits of the folded bitmap in LDS, log2 (15..20; < 20 also stages
	                                        the survivors' coordinates) */
	int32_t disable;                     /* bit mask NWAYHIP_DISABLE_* */
	int32_t enable;                      /* bit mask NWAYHIP_ENABLE_*: variants that are off by default */
} nwayhip_match_params;
#define NWAYHIP_DISABLE_DENSE3 1         /* dense k = 3: the hybrid path instead of the tuple-parallel fused tail */
#define NWAYHIP_DISABLE_HYBRID 2         /* dense k >= 3: the general path instead of sparse front + general back end */
#define NWAYHIP_DISABLE_FUSED_CORRECTION 4 /* NWAYHIP_CORRECTION_CLI: k_correct behind the general back end instead of the fused tails */
#define NWAYHIP_DISABLE_ONE_SWEEP 8      /* sparse k >= 3: one sweep launch per secondary catalogue instead of one for all */
#define NWAYHIP_DISABLE_QUAD3 16         /* sparse k = 3: k_tailk<3> (one lane per primary) instead of k_tail3q (four) */
/* (1: unused since round 4 -- round 3's registration inside the sweep launch, measured no faster and removed) */
#define NWAYHIP_ENABLE_QUAD3 2           /* sparse k = 3: k_tail3q whatever the density of chance neighbours (default: below 0.02 per primary) */

/* Output table, SoA, `capacity` rows allocated by the caller.  Columns follow
 * __init__.py:133-177,100-111,405-418 / SURVEY.md appendix C. */
typedef struct nwayhip_table {
	int64_t capacity;
	int32_t* idx[NWAYHIP_MAXCAT];        /* row index into catalogue c, -1 = absent */
	double* sep[NWAYHIP_MAXPAIR];        /* Separation_{i}_{j}, i<j, order (0,1),(0,2)..(1,2)..; NaN if absent */
	double* sep_max;                     /* Separation_max */
	int8_t* ncat;
	double* log_bf;                      /* dist_bayesfactor_uncorrected */
	double* log_bf_corrected;            /* dist_bayesfactor (== log_bf unless correction CLI); may be NULL */
	double* prior;                       /* may be NULL where nwayhip_plan_path() says so */
	double* dist_post;
	double* p_single;                    /* finalize only */
	double* p_any;                       /* prob_has_match */
	double* p_i;                         /* prob_this_match */
	int8_t* match_flag;
	int64_t* group_start;                /* [n_primary + 1] first row of every primary */
} nwayhip_table;

typedef struct nwayhip_plan nwayhip_plan;

/* ---- library ---------------------------------------------------------------------- */
int nwayhip_version(void);
const char* nwayhip_last_error(void);
int nwayhip_device_count(int* h_count);

/* ---- elementwise mirrors of the reference's array functions ------------------------ */
/* fastskymatch.py:26-47  dist(apos, bpos) -> degrees */
int nwayhip_dist(const double* a_ra, const double* a_dec, const double* b_ra, const double* b_dec,
	int64_t n, double* out_deg, void* stream);
/* the same for float32 positions: numpy keeps float32 through every operation and returns float32
 * (what the reference's match_multiple gets from FITS 'E' coordinate columns) */
int nwayhip_dist_f32(const float* a_ra, const float* a_dec, const float* b_ra, const float* b_dec,
	int64_t n, float* out_deg, void* stream);
/* bayesdistance.py:64-86  log_bf(p, s).  h_sep: host array of ncat*ncat device pointers
 * (row-major, only i<j read), h_sigma: host array of ncat device pointers. */
int nwayhip_log_bf(int32_t ncat, int64_t n, const double* const* h_sep, const double* const* h_sigma,
	double* out, void* stream);
/* bayesdistance.py:207-240  log_bf_elliptical(separations_ra, separations_dec, pos_errors): per-axis
 * separations (arcsec; host arrays of ncat*ncat device pointers, only i<j read) and, per catalogue,
 * the error ellipse as (sigma_x, sigma_y, rho) columns (host arrays of ncat device pointers).
 * f32_offsets = 1: the numerics of the script's main pass (nway.py:346-354), whose offsets are float32
 * arrays read back from FITS 'E' columns: numpy keeps their length and unit vector in float32. */
int nwayhip_log_bf_elliptical(int32_t ncat, int64_t n, const double* const* h_sep_ra, const double* const* h_sep_dec,
	const double* const* h_sigma_x, const double* const* h_sigma_y, const double* const* h_rho, double* out, int32_t f32_offsets,
	void* stream);
/* fastskymatch.py:50-74  the two offset columns of dist3d(apos, bpos): longitude and latitude
 * differences (degrees, a minus b) in the offset frame centred on a; -99 inputs give NaN.
 * (The separation column of dist3d is nwayhip_dist.) */
int nwayhip_offsets(const double* a_ra, const double* a_dec, const double* b_ra, const double* b_dec, int64_t n,
	double* d_lon_deg, double* d_lat_deg, void* stream);
/* bayesdistance.py:26-32 (mode 0: posterior), :18-23 (mode 1: log_posterior),
 * :35-39 (mode 2: unnormalised_log_posterior) */
int nwayhip_posterior(int32_t mode, const double* prior, const double* log_bf, int64_t n, double* out, void* stream);
/* Test hook, no reference counterpart: the elementary functions the row kernels evaluate in place of the device library's
 * (csrc/fastmath.inc: short roads for the arguments a match has), one call per element, so that they can be compared
 * bit for bit with the same source compiled for the host and with numpy (numpy.sin, arctan2, hypot, log, log10, 10**x of
 * fastskymatch.py:26-47 and bayesdistance.py:18-86).  fn: 0 sincos(x) -> out, out2; 1 atan2(x, y); 2 hypot(x, y);
 * 3 log(x); 4 log10(x); 5 10^x; 6 x / 180 * pi; 7 x * 180 / pi (the divisions by a literal, in three operations, correctly rounded).
 * y: only fn 1, 2; out2: only fn 0. */
int nwayhip_fastmath_probe(int32_t fn, const double* x, const double* y, int64_t n, double* out, double* out2, void* stream);

/* ---- the match pipeline ------------------------------------------------------------ */
/* Replaces crossproduct (fastskymatch.py:92-218), _create_match_table (__init__.py:123-196),
 * _compute_single_log_bf (:220-259), posterior (:110) and, with finalize,
 * _compute_final_probabilities (:399-461).
 *
 * A plan fixes the parameters, catalogue sizes and buffer capacities and carves the
 * caller's workspace; it owns no device memory. */
int nwayhip_plan_create(nwayhip_plan** plan, const nwayhip_match_params* params, const int64_t* h_n,
	int64_t cap_pairs, int64_t cap_rows);
int nwayhip_plan_destroy(nwayhip_plan* plan);
size_t nwayhip_plan_workspace_bytes(const nwayhip_plan* plan);
/* slots of the plan's cell table (what to multiply when NWAYHIP_FLAG_REG_OVERFLOW comes back) */
int64_t nwayhip_plan_table_slots(const nwayhip_plan* plan);
/* link slots per primary the plan settled on: > 0 = the sparse front runs (direct-mapped cell table,
 * probe inside the sweep, candidates in fixed slots of their primaries), 0 = the general path. */
int32_t nwayhip_plan_link_slots(const nwayhip_plan* plan);
/* which kernels the plan runs.  SPARSE: the sparse front and a fused tail; with finalize the
 * table's `prior` column may then be NULL (nothing reads it afterwards).  HYBRID (k >= 4 with tens
 * of links per primary): the sparse front feeding the general back end.  log_bf_corrected may be NULL on every path unless the correction is
 * NWAYHIP_CORRECTION_CLI (it equals log_bf). */
#define NWAYHIP_PATH_GENERAL 0
#define NWAYHIP_PATH_SPARSE 1
#define NWAYHIP_PATH_HYBRID 2
int32_t nwayhip_plan_path(const nwayhip_plan* plan);
/* What exactly the plan launches (tests of the path selection, configuration tables):
 * h_out[NWAYHIP_DESC_WORDS], indexed by NWAYHIP_DESC_*. */
#define NWAYHIP_DESC_WORDS 8
#define NWAYHIP_DESC_PATH 0              /* NWAYHIP_PATH_* */
#define NWAYHIP_DESC_LINK_SLOTS 1        /* slots per primary and catalogue (0: general path) */
#define NWAYHIP_DESC_DIRECT_LOG2 2       /* positions of the direct-mapped table, log2 (0: general path) */
#define NWAYHIP_DESC_SWEEP 3             /* NWAYHIP_SWEEP_* */
#define NWAYHIP_DESC_TAIL 4              /* NWAYHIP_TAIL_* */
#define NWAYHIP_DESC_FOLD_LOG2 5         /* large-table sweep: bits of the folded bitmap, log2 (else 0) */
#define NWAYHIP_DESC_ONE_SWEEP 6         /* 1: all secondary catalogues share one sweep launch */
/* (word 7: always 0 since round 4) */
#define NWAYHIP_SWEEP_GENERAL 0          /* survivors to regions, k_pairs + k_links behind it */
#define NWAYHIP_SWEEP_LDS 1              /* sparse front, occupancy bitmap in LDS (tables up to 2^20 positions) */
#define NWAYHIP_SWEEP_BIG 2              /* sparse front, bitmap in L2, folded copy in LDS */
#define NWAYHIP_TAIL_GENERAL 0           /* lists + k_rows2 / expansion + k_rows + k_groups */
#define NWAYHIP_TAIL_SPARSE2 1           /* k_tail2 */
#define NWAYHIP_TAIL_DENSE2 2            /* k_taild_test + scan + k_taild_rows */
#define NWAYHIP_TAIL_SPARSEK 3           /* k_tailk<K> */
#define NWAYHIP_TAIL_DENSE3 4            /* k_taild_test + k_taild3_count + scan + k_taild3_rows */
#define NWAYHIP_TAIL_HYBRID 5            /* slots -> lists, then the general back end */
#define NWAYHIP_TAIL_QUAD3 6             /* k_tail3q: k = 3, four lanes per primary */
int nwayhip_plan_describe(const nwayhip_plan* plan, int32_t* h_out);
/* 1 if the plan can run the secondary-split mode below (sparse front with the tails
 * NWAYHIP_TAIL_SPARSE2 / NWAYHIP_TAIL_DENSE2 / NWAYHIP_TAIL_SPARSEK / NWAYHIP_TAIL_QUAD3), else 0.  With NWAYHIP_TAIL_QUAD3 a
 * run may come back with NWAYHIP_FLAG_QUAD_DEEP: every caller of the split entry points repeats it with
 * NWAYHIP_DISABLE_QUAD3, as nway_amd/distributed.py does */
int32_t nwayhip_plan_split_capable(const nwayhip_plan* plan);
/* Enqueue the whole pipeline on `stream`.  d_status: device int64[NWAYHIP_STATUS_WORDS].
 * The workspace's contents may be arbitrary the first time a plan sees it (the plan clears what
 * it needs); between runs of the same plan on the same workspace they must be left alone (by
 * the caller and by other plans) -- the cell table is epoch-tagged and not cleared again.  Handing the plan a different workspace
 * pointer is always safe; after scribbling over a workspace, destroy the plan and make a new one. */
int nwayhip_match_enqueue(nwayhip_plan* plan, const nwayhip_catalogue* h_cats, void* workspace,
	size_t workspace_bytes, const nwayhip_table* h_table, int64_t* d_status, void* stream);

/* ---- several zones of one job as ONE launch set -------------------------------------------------
 * (The reference fills ONE bucket table per job, fastskymatch.py:118-160; a job whose table outgrows the LDS of a sweep
 * workgroup is cut into declination zones, each an ordinary plan of its own -- nway_amd/distributed.py: ZoneShardedMatch.)
 * Run one by one, Z zones pay the fixed latencies of a pass Z times.  A zones object holds the plans of the zones and
 * enqueues ALL their registrations as one launch, all their sweeps as one launch, all their tails as one launch: the
 * workgroups of a launch are shared out among the zones (the sweep's in proportion to the sources a zone streams), every
 * zone's argument block lives in `d_args` (device memory of the caller, nwayhip_zones_args_bytes(), 256-byte aligned;
 * written by the library with a stream-ordered copy whenever a block changes -- catalogue or table pointers, not every run).
 * Every zone keeps its own workspace, table and status block: results and status words are exactly those of the zones'
 * own nwayhip_match_enqueue calls.  Batched where every plan takes the sparse front with the bitmap in LDS and the 2-way
 * sparse tail (NWAYHIP_SWEEP_LDS + NWAYHIP_TAIL_SPARSE2) and has run once on its workspace (its first run clears it);
 * otherwise -- and always correctly -- the zones are enqueued one after the other.  nwayhip_zones_batched(): which of the
 * two the last enqueue was.  The plans must outlive the object; nwayhip_plan_profile on the FIRST plan brackets the
 * launches of the whole set. */
typedef struct nwayhip_zones nwayhip_zones;
typedef struct nwayhip_zone_run {
	const nwayhip_catalogue* h_cats;     /* as nwayhip_match_enqueue, per zone */
	void* workspace;
	size_t workspace_bytes;
	const nwayhip_table* h_table;
	int64_t* d_status;
} nwayhip_zone_run;
#define NWAYHIP_MAXZONES 64
int nwayhip_zones_create(nwayhip_zones** zones, nwayhip_plan* const* h_plans, int32_t nplans);
int nwayhip_zones_destroy(nwayhip_zones* zones);
size_t nwayhip_zones_args_bytes(const nwayhip_zones* zones);
int nwayhip_zones_enqueue(nwayhip_zones* zones, const nwayhip_zone_run* h_runs /*[nplans]*/, void* d_args, size_t d_args_bytes, void* stream);
/* 0: the last enqueue went out zone by zone; 1: as one launch set; 2: as one launch set whose registration was owner-computes */
int32_t nwayhip_zones_batched(const nwayhip_zones* zones);
/* How a launch set registers its primaries in their cell tables (fastskymatch.py:118-133, "only the primary catalogue is allowed
 * to define new buckets").  ATOMICS: every registration claims its table position with an atomic in memory, as a plan of its own
 * does.  OWNER: no atomic per registration -- the registrations are sorted into the slices of their tables and every slice has one
 * workgroup that claims in LDS (two launches; csrc/zones.inc) -- what pays once the atomics are a throughput: AUTO (the default)
 * takes it from 200 000 primaries per launch set on.  The tables obey the same invariants either way, the results are the same. */
#define NWAYHIP_ZONES_REG_AUTO 0
#define NWAYHIP_ZONES_REG_ATOMICS 1
#define NWAYHIP_ZONES_REG_OWNER 2
int nwayhip_zones_set_registration(nwayhip_zones* zones, int32_t mode);

/* ---- secondary-split mode: several GPUs on ONE job ---------------------------------------
 * (second half; the reference is a single process.)  Every rank registers ALL
 * primaries and sweeps its own slice of every secondary catalogue; a candidate (primary,
 * secondary) is exported to the rank that owns the primary; one all-to-all of the export buffers
 * (the caller's: torch.distributed / RCCL, nway_amd/distributed.py) delivers them; the back half
 * turns what arrived into the links of the rank's own primaries and runs the fused tail on them.
 * Only where nwayhip_plan_split_capable() says so.  The plan is created with n[0] = ALL primaries
 * and n[c] = the rank's slice sizes; capacities and the table are for the rank's own rows.
 *
 * Export buffer: [destination rank][catalogue c - 1][1 + capacity] records of 32 bytes
 * {int32 primary, int32 secondary (global indices), double ra, dec, sigma}; record 0 of a block
 * is its header (`primary` = number of records).  Headers must be zero before the first front
 * half; the back half resets them.  More records than `capacity` for one peer: the receiver
 * raises NWAYHIP_FLAG_PAIR_OVERFLOW (come back with a larger capacity). */
typedef struct nwayhip_split {
	int32_t world, rank;
	const int64_t* d_bounds;             /* device int64[world + 1]: rank r owns primaries [bounds[r], bounds[r+1]) */
	int64_t h_p_lo, h_p_hi;              /* = bounds[rank], bounds[rank + 1] */
	int64_t slice_offset[NWAYHIP_MAXCAT];/* [c >= 1]: global index of the first row of this rank's slice of catalogue c */
	int64_t capacity;                    /* records per (peer, catalogue) block */
	void* d_export;                      /* nwayhip_split_buffer_bytes() */
	const void* d_import;                /* what the all-to-all delivered (same layout, [source rank] first) */
} nwayhip_split;
size_t nwayhip_split_buffer_bytes(const nwayhip_plan* plan, int32_t world, int64_t capacity);
/* register (all primaries) + sweep (own slices) -> export buffer */
int nwayhip_split_front_enqueue(nwayhip_plan* plan, const nwayhip_catalogue* h_cats, void* workspace, size_t workspace_bytes,
	const nwayhip_split* h_split, int64_t* d_status, void* stream);
/* import buffer -> links of the own primaries -> fused tail -> table (rows of the own primaries, global indices) */
int nwayhip_split_back_enqueue(nwayhip_plan* plan, const nwayhip_catalogue* h_cats, void* workspace, size_t workspace_bytes,
	const nwayhip_split* h_split, const nwayhip_table* h_table, int64_t* d_status, void* stream);

/* ---- RCCL behind the ABI: the two exchanges of the multi-GPU modes as library calls ------
 * One process per GPU.  Rank 0 makes the id, the host hands its NWAYHIP_COMM_ID_BYTES bytes to every rank
 * (any side channel), every rank calls nwayhip_comm_init with its device current; the calls are collective.
 * librccl is opened when the first of these is called (a process that never shards does not need it).
 * nway_amd/distributed.py takes this route with comm='rccl'; its default remains torch.distributed. */
#define NWAYHIP_COMM_ID_BYTES 128
typedef struct nwayhip_comm nwayhip_comm;
int nwayhip_comm_unique_id(void* h_id /*[NWAYHIP_COMM_ID_BYTES]*/);
int nwayhip_comm_init(nwayhip_comm** comm, int32_t world, int32_t rank, const void* h_id);
int nwayhip_comm_destroy(nwayhip_comm* comm);
int32_t nwayhip_comm_world(const nwayhip_comm* comm);
int32_t nwayhip_comm_rank(const nwayhip_comm* comm);
/* all-gatherv of one double column: rank r contributes h_counts[r] rows (d_send), d_recv receives the sum(h_counts)
 * rows of all ranks in rank order -- one group of broadcasts on `stream` */
int nwayhip_comm_allgatherv_f64(nwayhip_comm* comm, const double* d_send, const int64_t* h_counts /*[world]*/, double* d_recv, void* stream);
/* the candidate exchange of the secondary-split mode: block r (block_bytes bytes) of d_export to rank r, block s of
 * d_import from rank s; one group of send / recv pairs on `stream`, between the two halves below, no host round trip */
int nwayhip_comm_exchange(nwayhip_comm* comm, const void* d_export, void* d_import, size_t block_bytes, void* stream);

/* Stage timing with HIP events recorded on the pipeline's own stream (bench.py's roofline leg).
 * stage_mask: bit s set => every launch group of stage s in subsequent nwayhip_match_enqueue
 * calls is bracketed by an event pair (ring of NWAYHIP_PROFILE_RING pairs per stage); the
 * sweep, a single launch, carries its pair on its own dispatch (kernel begin / end timestamps).
 * nwayhip_plan_profile_read waits for the recorded events, returns per stage the number of
 * bracketed launch groups and their summed duration in ms, and resets the counters. */
#define NWAYHIP_STAGE_REGISTER 0
#define NWAYHIP_STAGE_SWEEP 1
#define NWAYHIP_STAGE_PAIRS 2
#define NWAYHIP_STAGE_LISTS 3
#define NWAYHIP_STAGE_EXPAND 4
#define NWAYHIP_STAGE_ROWS 5
#define NWAYHIP_STAGE_GROUPS 6
#define NWAYHIP_STAGES 8
#define NWAYHIP_PROFILE_RING 512
int nwayhip_plan_profile(nwayhip_plan* plan, uint32_t stage_mask);
int nwayhip_plan_profile_read(nwayhip_plan* plan, int64_t* h_launches /*[NWAYHIP_STAGES]*/, double* h_ms /*[NWAYHIP_STAGES]*/);
/* The individual durations (ms) of the bracketed launch groups of one stage recorded so far, at most `capacity` of them
 * (bench.py: minimum / median / maximum of the sweep's launches); waits for their events, resets nothing -- call it before
 * nwayhip_plan_profile_read. */
int nwayhip_plan_profile_samples(nwayhip_plan* plan, int32_t stage, double* h_ms /*[capacity]*/, int64_t capacity, int64_t* h_count);
/* An event pair on a dispatch costs the pipeline a few microseconds on this stack: with `every` > 1
 * only every `every`-th launch of a single-launch stage (the sweep) carries one.  Default 1. */
int nwayhip_plan_profile_stride(nwayhip_plan* plan, int32_t every);

/* Per-primary group statistics (__init__.py:399-461 == nway.py:527-586) on a table whose
 * total = log_bf + sum(biases) was assembled by the caller (magnitude priors).
 * group_start: int64[n_groups + 1].  Writes p_single, p_any, p_i, match_flag. */
int nwayhip_group_stats(int64_t n_rows, int64_t n_groups, const int64_t* group_start, const double* total,
	const double* prior, double prob_ratio_secondary, double* p_single, double* p_any, double* p_i,
	int8_t* match_flag, void* stream);

/* Magnitude bias of one column (__init__.py:383-394 with magnitudeweights.fitfunc_histogram
 * :74-87): row r takes mag[idx[r]] (idx < 0 or undefined magnitude -> weight 0), looks it up in
 * the step function ratio[bin] over edges[0..n_edges-1] (ratio has n_edges-1 entries),
 * weight = log10(ratio); total_inout[r] += weight; bias_out[r] = 10^weight. */
int nwayhip_bias_lookup(int64_t n_rows, const int32_t* idx, const double* mag, int32_t n_edges, const double* edges,
	const double* ratio, double* total_inout, double* bias_out, void* stream);

/* Measurement aid (no counterpart in the reference): reads two double columns of n rows once with the
 * sweep's access pattern -- `blocks` workgroups of 1024 threads on contiguous slices -- and leaves one
 * partial sum per workgroup in d_out[blocks].  bench.py times it as the machine's ceiling for the
 * sweep's 16 bytes per secondary (roofline.read_peak). */
int nwayhip_read_probe(const double* a, const double* b, int64_t n, double* d_out, int32_t blocks, void* stream);

/* fastskymatch.py:94-98 on device-resident columns: out[0..3] = min ra, max ra, max |dec|, #NaN */
int nwayhip_catalogue_extent(const double* ra, const double* dec, int64_t n, double* d_out4, void* stream);

#ifdef __cplusplus
}
#endif
#endif
