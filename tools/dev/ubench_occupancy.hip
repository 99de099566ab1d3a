// development aid: how many workgroups of a small kernel a CU of this part actually holds at once (round 6: the registration kernels of a launch set
// ran as rounds of 1 024 threads per CU whatever their registers and LDS).  Every workgroup stamps its start, idles for ~15 us, stamps its end; the host
// counts the workgroups that had started when the first one ended.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/ubench_occupancy.hip -o gpurun_out/ubench_occupancy && gpurun_out/ubench_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_idle(long long* stamps, int lds_words, long long ticks) {
	extern __shared__ int lds[];
	if (lds_words > 0 && threadIdx.x < 32) lds[threadIdx.x % lds_words] = threadIdx.x;
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0) {
		stamps[2 * blockIdx.x] = t0;
		stamps[2 * blockIdx.x + 1] = wall_clock64();
	}
}

template <int THREADS>
void run(int blocks, int lds_bytes) {
	long long* d;
	CHECK(hipMalloc(&d, sizeof(long long) * 2 * blocks));
	CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_idle<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	for (int rep = 0; rep < 2; ++rep) {
		hipLaunchKernelGGL(k_idle<THREADS>, dim3(blocks), dim3(THREADS), lds_bytes, 0, d, lds_bytes / 4, 1500ll);  // 100 MHz: 15 us
		CHECK(hipDeviceSynchronize());
	}
	std::vector<long long> h(2 * blocks);
	CHECK(hipMemcpy(h.data(), d, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost));
	long long first_start = h[0], first_end = h[1];
	for (int b = 0; b < blocks; ++b) {
		first_start = std::min(first_start, h[2 * b]);
		first_end = std::min(first_end, h[2 * b + 1]);
	}
	int early = 0;
	for (int b = 0; b < blocks; ++b) early += h[2 * b] < first_end ? 1 : 0;
	long long last_end = 0;
	for (int b = 0; b < blocks; ++b) last_end = std::max(last_end, h[2 * b + 1]);
	printf("%4d threads, %6d B LDS, %5d workgroups: %5d resident at once = %.1f waves per CU (256 CUs); whole launch %.1f us for an idle of 15\n", THREADS, lds_bytes, blocks, early,
		early * (THREADS / 64) / 256.0, (last_end - first_start) * 0.01);
	CHECK(hipFree(d));
}

int main() {
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	printf("%s: %d CUs, maxThreadsPerMultiProcessor %d, sharedMemPerMultiprocessor %zu, regsPerMultiprocessor %d\n", prop.gcnArchName, prop.multiProcessorCount,
		prop.maxThreadsPerMultiProcessor, (size_t)prop.sharedMemPerMultiprocessor, prop.regsPerMultiprocessor);
	run<64>(16384, 0);
	run<256>(8192, 0);
	run<256>(8192, 16 * 1024);
	run<256>(8192, 32 * 1024);
	run<512>(4096, 0);
	run<1024>(2048, 0);
	run<1024>(2048, 64 * 1024);
	return 0;
}
