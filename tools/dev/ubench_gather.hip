// development aid: the rate of scattered 4-byte gathers on gfx950 against the size of the table
// they hit -- what bounds the large-table sweep (nway_amd/csrc/sweepbig.inc): one lane,
// one line per gather ("transaction"), alone and beside a coalesced 16-byte-per-lane stream
//   hipcc --offload-arch=gfx950 -O3 tools/dev/ubench_gather.hip -o gpurun_out/ubench_gather && gpurun_out/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}

typedef double dbl2 __attribute__((ext_vector_type(2)));

// every lane: `per_lane` rounds of (optionally) one nontemporal 16-byte load of the stream and
// `gathers` dependent-free 4-byte gathers from `table` (words a power of two)
template <int GATHERS, bool STREAM>
__global__ void __launch_bounds__(1024) k_gather(const uint32_t* __restrict__ table, uint32_t mask, const dbl2* __restrict__ stream, long long nvec,
	int per_lane, uint32_t* sink) {
	const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (long long)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	for (int r = 0; r < per_lane; ++r) {
		uint32_t h = mix((uint32_t)t * 2654435761u + (uint32_t)r * 40503u);
		if (STREAM) {
			const long long v = ((long long)r * nthreads + t) % nvec;
			const dbl2 x = __builtin_nontemporal_load(&stream[v]);
			h ^= (uint32_t)__double_as_longlong(x.x) ^ (uint32_t)__double_as_longlong(x.y);
		}
#pragma unroll
		for (int g = 0; g < GATHERS; ++g) acc += table[(h + g * 0x9E3779B9u) & mask];
	}
	if (acc == 0x12345678u) *sink = acc;
}

template <int GATHERS, bool STREAM>
static double run(const uint32_t* table, uint32_t words, const dbl2* stream, long long nvec, int per_lane, uint32_t* sink) {
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int blocks = 256, threads = 1024;
	hipLaunchKernelGGL((k_gather<GATHERS, STREAM>), dim3(blocks), dim3(threads), 0, 0, table, words - 1, stream, nvec, per_lane, sink);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	const int reps = 5;
	for (int i = 0; i < reps; ++i)
		hipLaunchKernelGGL((k_gather<GATHERS, STREAM>), dim3(blocks), dim3(threads), 0, 0, table, words - 1, stream, nvec, per_lane, sink);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	return ms * 1e3 / reps;  // us per launch
}

int main() {
	const long long nvec = 50000000;  // 800 MB stream of 16-byte vectors
	dbl2* stream;
	uint32_t *table, *sink;
	CHECK(hipMalloc(&stream, nvec * sizeof(dbl2)));
	CHECK(hipMemset(stream, 0, nvec * sizeof(dbl2)));
	CHECK(hipMalloc(&table, (size_t)256 << 20));
	CHECK(hipMemset(table, 0, (size_t)256 << 20));
	CHECK(hipMalloc(&sink, 4));
	const int per_lane = 190;  // 256 x 1024 lanes x 190 = 5e7 rounds
	const double rounds = 256.0 * 1024 * per_lane;
	printf("%-10s | %-28s | %-28s | %-28s\n", "table", "1 gather / round", "2 gathers / round", "1 gather + 16-byte stream load / round");
	for (int lg = 17; lg <= 28; lg += (lg < 24 ? 1 : 2)) {  // bytes: 128 KiB .. 256 MiB
		const uint32_t words = 1u << (lg - 2);
		const double a = run<1, false>(table, words, stream, nvec, per_lane, sink);
		const double b = run<2, false>(table, words, stream, nvec, per_lane, sink);
		const double c = run<1, true>(table, words, stream, nvec, per_lane, sink);
		printf("%6.1f MiB | %8.1f us %6.1f gathers/ns | %8.1f us %6.1f gathers/ns | %8.1f us %6.1f gathers/ns %5.2f TB/s\n", (double)(1u << lg) / (1 << 20),
			a, rounds / a * 1e-3, b, 2 * rounds / b * 1e-3, c, rounds / c * 1e-3, rounds * 16 / c * 1e-6);
	}
	return 0;
}
