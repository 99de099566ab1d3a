// development aid: what the HBM of this part gives a kernel that reads two columns once (the sweep's stream:
// nway_amd/csrc/front.inc, elementwise.inc: k_read_probe), against the way the tiles are handed to the workgroups,
// the workgroup size, the tiles in flight and the load flavour.  Three sets of columns alternate so that no launch
// finds its 160 MB in the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/ubench_read.hip -o gpurun_out/ubench_read && gpurun_out/ubench_read
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double dbl2 __attribute__((ext_vector_type(2)));

// MODE 0: contiguous slice per workgroup (the sweep's)      MODE 1: tile t goes to workgroup t % grid
// DEPTH tiles of THREADS vectors per column in flight; NT: nontemporal loads
template <int THREADS, int MODE, int DEPTH, bool NT>
__global__ void __launch_bounds__(THREADS) k_read(const dbl2* __restrict__ a, const dbl2* __restrict__ b, long long nvec, double* __restrict__ out) {
	dbl2 acc = {0.0, 0.0};
	const long long ntiles = (nvec + THREADS - 1) / THREADS;
	long long t0, t1, step;
	if (MODE == 0) {
		const long long per = (ntiles + gridDim.x - 1) / gridDim.x;
		t0 = (long long)blockIdx.x * per;
		t1 = min(ntiles, t0 + per);
		step = 1;
	} else {
		t0 = blockIdx.x;
		t1 = ntiles;
		step = gridDim.x;
	}
	for (long long t = t0; t < t1; t += step * DEPTH) {
		dbl2 x[DEPTH], y[DEPTH];
#pragma unroll
		for (int d = 0; d < DEPTH; ++d) {
			long long v = (t + d * step) * THREADS + threadIdx.x;
			if (t + d * step >= t1 || v >= nvec) v = nvec - 1;
			x[d] = NT ? __builtin_nontemporal_load(&a[v]) : a[v];
			y[d] = NT ? __builtin_nontemporal_load(&b[v]) : b[v];
		}
#pragma unroll
		for (int d = 0; d < DEPTH; ++d) acc += x[d] + y[d];
	}
	if (acc.x + acc.y == 0.123456789) out[blockIdx.x] = acc.x;
}

template <int THREADS, int MODE, int DEPTH, bool NT>
static void run(const char* what, dbl2* const* a, dbl2* const* b, long long nvec, int blocks, double* out) {
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_read<THREADS, MODE, DEPTH, NT>), dim3(blocks), dim3(THREADS), 0, 0, a[i], b[i], nvec, out);
	CHECK(hipDeviceSynchronize());
	const int reps = 12;
	CHECK(hipEventRecord(e0));
	for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_read<THREADS, MODE, DEPTH, NT>), dim3(blocks), dim3(THREADS), 0, 0, a[i % 3], b[i % 3], nvec, out);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	const double us = ms * 1e3 / reps;
	printf("%-64s %4d workgroups of %4d  %7.2f us  %6.3f TB/s\n", what, blocks, THREADS, us, 32.0 * nvec / us * 1e-6);
}

// the library's own probe (elementwise.inc: k_read_probe) on THIS program's buffers, timed like the kernels above
static void run_library(const char* path, dbl2* const* a, dbl2* const* b, long long nvec, int blocks, double* out) {
	void* h = dlopen(path, RTLD_NOW);
	if (!h) { printf("dlopen %s: %s\n", path, dlerror()); return; }
	typedef int (*probe_fn)(const double*, const double*, long long, double*, int, void*);
	probe_fn f = (probe_fn)dlsym(h, "nwayhip_read_probe");
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int i = 0; i < 3; ++i) f((const double*)a[i], (const double*)b[i], 2 * nvec, out, blocks, nullptr);
	CHECK(hipDeviceSynchronize());
	const int reps = 12;
	CHECK(hipEventRecord(e0));
	for (int i = 0; i < reps; ++i) f((const double*)a[i % 3], (const double*)b[i % 3], 2 * nvec, out, blocks, nullptr);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	const double us = ms * 1e3 / reps;
	printf("%-64s %4d workgroups of %4d  %7.2f us  %6.3f TB/s\n", "libnwayhip.so: nwayhip_read_probe (nontemporal)", blocks, 1024, us, 32.0 * nvec / us * 1e-6);
}

int main(int argc, char** argv) {
	const long long nvec = 5000000;  // 2 x 80 MB
	dbl2 *a[3], *b[3];
	for (int i = 0; i < 3; ++i) {
		CHECK(hipMalloc(&a[i], nvec * 16));
		CHECK(hipMalloc(&b[i], nvec * 16));
		CHECK(hipMemset(a[i], 0, nvec * 16));
		CHECK(hipMemset(b[i], 0, nvec * 16));
	}
	double* out;
	CHECK(hipMalloc(&out, 4096 * 8));
	run<1024, 0, 2, false>("contiguous slices, 2 tiles in flight (the sweep, k_read_probe)", a, b, nvec, 256, out);
	run<1024, 0, 2, true>("contiguous slices, 2 tiles, nontemporal", a, b, nvec, 256, out);
	if (argc > 1) run_library(argv[1], a, b, nvec, 256, out);
	run<1024, 0, 2, true>("contiguous slices, 2 tiles, nontemporal (again)", a, b, nvec, 256, out);
	run<1024, 0, 4, false>("contiguous slices, 4 tiles in flight", a, b, nvec, 256, out);
	run<1024, 0, 1, false>("contiguous slices, 1 tile in flight", a, b, nvec, 256, out);
	run<1024, 1, 2, false>("tiles round robin over the workgroups, 2 in flight", a, b, nvec, 256, out);
	run<1024, 1, 4, false>("tiles round robin, 4 in flight", a, b, nvec, 256, out);
	run<1024, 1, 2, true>("tiles round robin, 2 in flight, nontemporal", a, b, nvec, 256, out);
	run<512, 0, 2, false>("contiguous slices, 2 tiles", a, b, nvec, 512, out);
	run<512, 1, 2, false>("tiles round robin, 2 tiles", a, b, nvec, 512, out);
	run<256, 0, 4, false>("contiguous slices, 4 tiles", a, b, nvec, 1024, out);
	run<256, 1, 4, false>("tiles round robin, 4 tiles", a, b, nvec, 1024, out);
	run<256, 1, 4, false>("tiles round robin, 4 tiles", a, b, nvec, 2048, out);
	run<256, 1, 2, false>("tiles round robin, 2 tiles", a, b, nvec, 4096, out);
	run<256, 0, 2, false>("contiguous slices, 2 tiles", a, b, nvec, 4096, out);
	return 0;
}
