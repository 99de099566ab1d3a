// development aid: what a dependent FP64 chain costs on this part, against the waves per SIMD and the
// independent chains per lane -- the question behind the one-lane-per-primary tails (tail2.inc): is a wave
// alone on its SIMD bound by the ISSUE of its FP64 instructions (4 cycles per wave64 instruction on the
// 16-lane FP64 pipe) or by their LATENCY, i.e. does a second wave on the SIMD (or a second independent
// chain in the lane) come for free?
//   hipcc --offload-arch=gfx950 -O3 tools/dev/ubench_fp64.hip -o gpurun_out/ubench_fp64 && gpurun_out/ubench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// K independent chains of N dependent FMAs each
template <int K>
__global__ void __launch_bounds__(64) k_fma(double* out, int n, double b, double c) {
	double a[K];
#pragma unroll
	for (int k = 0; k < K; ++k) a[k] = (double)(threadIdx.x + k) * 1e-3;
	for (int i = 0; i < n; ++i) {
#pragma unroll
		for (int k = 0; k < K; ++k) a[k] = __builtin_fma(a[k], b, c);
	}
	double s = 0;
#pragma unroll
	for (int k = 0; k < K; ++k) s += a[k];
	if (s == 0.123456789) out[blockIdx.x] = s;
}

// the tails' own mix: K independent evaluations of sincos -> atan2(hypot) -> log -> exp10 per lane
template <int K>
__global__ void __launch_bounds__(64) k_libm(double* out, int n, double x0) {
	double acc = 0;
	for (int i = 0; i < n; ++i) {
		double v[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const double x = x0 + 1e-3 * (threadIdx.x + 64 * k + i);
			double s, c;
			sincos(x, &s, &c);
			const double t = atan2(hypot(s, 0.5 * c), c + 1.5);
			v[k] = exp10(-log(t + 1.0));
		}
#pragma unroll
		for (int k = 0; k < K; ++k) acc += v[k];
	}
	if (acc == 0.123456789) out[blockIdx.x] = acc;
}

template <class F>
static float time_it(F launch) {
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	launch();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	const int reps = 5;
	for (int i = 0; i < reps; ++i) launch();
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	return ms / reps;
}

int main() {
	double* out;
	CHECK(hipMalloc(&out, 1 << 20));
	const int n = 20000;
	printf("FMA chains: us for %d dependent v_fma_f64 per chain (cycles per FMA of a wave at 2.4 GHz)\n", n);
	printf("%8s %10s %10s %10s\n", "waves", "K=1", "K=2", "K=4");
	for (int waves : {256, 1024, 1536, 2048, 3072, 4096, 8192}) {
		const float t1 = time_it([&] { hipLaunchKernelGGL(k_fma<1>, dim3(waves), dim3(64), 0, 0, out, n, 1.0000001, 1e-9); });
		const float t2 = time_it([&] { hipLaunchKernelGGL(k_fma<2>, dim3(waves), dim3(64), 0, 0, out, n, 1.0000001, 1e-9); });
		const float t4 = time_it([&] { hipLaunchKernelGGL(k_fma<4>, dim3(waves), dim3(64), 0, 0, out, n, 1.0000001, 1e-9); });
		printf("%8d %7.1f (%4.1f) %7.1f (%4.1f) %7.1f (%4.1f)\n", waves, t1 * 1e3, t1 * 1e-3 * 2.4e9 / n, t2 * 1e3, t2 * 1e-3 * 2.4e9 / n / 2,
			t4 * 1e3, t4 * 1e-3 * 2.4e9 / n / 4);
	}
	const int m = 200;
	printf("libm mix (sincos, hypot, atan2, log, exp10): us for %d rounds; per evaluation in ns\n", m);
	printf("%8s %10s %10s %10s\n", "waves", "K=1", "K=2", "K=4");
	for (int waves : {256, 1024, 1536, 2048, 3072, 4096, 8192}) {
		const float t1 = time_it([&] { hipLaunchKernelGGL(k_libm<1>, dim3(waves), dim3(64), 0, 0, out, m, 0.3); });
		const float t2 = time_it([&] { hipLaunchKernelGGL(k_libm<2>, dim3(waves), dim3(64), 0, 0, out, m, 0.3); });
		const float t4 = time_it([&] { hipLaunchKernelGGL(k_libm<4>, dim3(waves), dim3(64), 0, 0, out, m, 0.3); });
		printf("%8d %7.1f (%5.0f) %7.1f (%5.0f) %7.1f (%5.0f)\n", waves, t1 * 1e3, t1 * 1e6 / m, t2 * 1e3, t2 * 1e6 / m / 2, t4 * 1e3, t4 * 1e6 / m / 4);
	}
	return 0;
}
