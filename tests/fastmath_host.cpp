// Host build of nway_amd/csrc/fastmath.inc for tests/test_fastmath_host.py and test_fastmath.py (built by tests/fastmath_util.py): the functions use IEEE operations
// only, so this is the arithmetic the kernels run (test infrastructure; not part of the library).
#include <cmath>
#define NW_FN static inline
#define NW_SLOW_FN static
#include "../nway_amd/csrc/fastmath.inc"
extern "C" {
void fm_sincos(const double* x, double* s, double* c, long n) { for (long i = 0; i < n; ++i) nw_sincos(x[i], &s[i], &c[i]); }
void fm_atan2(const double* y, const double* x, double* o, long n) { for (long i = 0; i < n; ++i) o[i] = nw_atan2(y[i], x[i]); }
void fm_hypot(const double* a, const double* b, double* o, long n) { for (long i = 0; i < n; ++i) o[i] = nw_hypot(a[i], b[i]); }
void fm_log(const double* x, double* o, long n) { for (long i = 0; i < n; ++i) o[i] = nw_log(x[i]); }
void fm_exp10(const double* x, double* o, long n) { for (long i = 0; i < n; ++i) o[i] = nw_exp10(x[i]); }
void fm_radians(const double* x, double* o, long n) { for (long i = 0; i < n; ++i) o[i] = nw_radians(x[i]); }
void fm_degrees(const double* x, double* o, long n) { for (long i = 0; i < n; ++i) o[i] = NW_DIV_K(x[i] * 180, M_PI); }
void fm_log10(const double* x, double* o, long n) { for (long i = 0; i < n; ++i) o[i] = nw_log10(x[i]); }
}
