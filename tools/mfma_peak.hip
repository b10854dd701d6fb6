// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 and v_mfma_f32_32x32x2_f32 on this GPU
// (register operands only, no memory traffic).  Used to state the MFMA peak that bench.py's roofline
// fraction is computed against.   hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

__device__ unsigned long long g_clk[4];
template <int NACC>
__global__ void __launch_bounds__(256) k_f64(double* out, int iters) {
	d4 acc[NACC];
	for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
	double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-3;
	unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
	}
	double s = 0;
	for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (blockIdx.x == 17 && threadIdx.x == 0) {
		g_clk[0] = t1 - t0;
		g_clk[1] = r1 - r0;
	}
}

template <int NACC>
__global__ void __launch_bounds__(256) k_f32(float* out, int iters) {
	f16 acc[NACC];
	for (int i = 0; i < NACC; i++)
		for (int j = 0; j < 16; j++) acc[i][j] = 0;
	float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0f - threadIdx.x * 1e-3f;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
	}
	float s = 0;
	for (int i = 0; i < NACC; i++)
		for (int j = 0; j < 16; j++) s += acc[i][j];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_valu64(double* out, int iters) {
	double acc[NACC];
	for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-9 + i;
	double a = threadIdx.x * 1e-3 + 1.0, b = 1e-7 * threadIdx.x;
	unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < NACC; i++) acc[i] = fma(acc[i], a, b);
	}
	double s = 0;
	for (int i = 0; i < NACC; i++) s += acc[i];
	unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (blockIdx.x == 17 && threadIdx.x == 0) {
		g_clk[0] = t1 - t0;
		g_clk[1] = r1 - r0;
	}
}

// co-execution probe: waves 0-3 issue f64 MFMA, waves 4-7 (their SIMD partners) issue v_fma_f64
__global__ void __launch_bounds__(512) k_mix(double* out, int iters_m, int iters_v) {
	double s = 0;
	if (threadIdx.x < 256) {
		d4 acc[8];
		for (int i = 0; i < 8; i++) acc[i] = (d4){0, 0, 0, 0};
		double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-3;
		for (int it = 0; it < iters_m; it++) {
#pragma unroll
			for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
		}
		for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	} else {
		double acc[16];
		for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-9 + i;
		double a = threadIdx.x * 1e-3 + 1.0, b = 1e-7 * threadIdx.x;
		for (int it = 0; it < iters_v; it++) {
#pragma unroll
			for (int i = 0; i < 16; i++) acc[i] = fma(acc[i], a, b);
		}
		for (int i = 0; i < 16; i++) s += acc[i];
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static void runmix(void* buf, int iters_m, int iters_v, hipEvent_t e0, hipEvent_t e1) {
	float ms = 0;
	for (int rep = 0; rep < 3; rep++) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, (double*)buf, iters_m, iters_v);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		(void)hipEventElapsedTime(&ms, e0, e1);
	}
	double fm = 256.0 * 4 * iters_m * 8 * 2048.0, fv = 256.0 * 256 * iters_v * 16 * 2.0;
	printf("mix          : mfma iters %d, valu iters %d: %.2f ms  -> MFMA %.1f + VALU %.1f = %.1f TFLOP/s\n", iters_m, iters_v, ms,
		   fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
}

template <int NACC>
static void runv(void* buf, int grid, int iters, hipEvent_t e0, hipEvent_t e1, const char* tag) {
	float ms = 0;
	for (int rep = 0; rep < 3; rep++) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k_valu64<NACC>, dim3(grid), dim3(256), 0, 0, (double*)buf, iters);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		(void)hipEventElapsedTime(&ms, e0, e1);
	}
	double fl = (double)grid * 256 * iters * NACC * 2.0;
	unsigned long long clk[4];
	(void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
	printf("v_fma_f64    : %s, %2d chains: %.2f ms  %.1f TFLOP/s   wave: %.2f cycles/FMA, clock %.0f MHz\n", tag, NACC, ms, fl / ms / 1e9,
		   (double)clk[0] / ((double)iters * NACC), 100.0 * (double)clk[0] / (double)clk[1]);
}

template <int NACC>
static void run64(void* buf, int grid, int iters, hipEvent_t e0, hipEvent_t e1, const char* tag) {
	float ms = 0;
	for (int rep = 0; rep < 3; rep++) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k_f64<NACC>, dim3(grid), dim3(256), 0, 0, (double*)buf, iters);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		(void)hipEventElapsedTime(&ms, e0, e1);
	}
	double fl = (double)grid * 4 * iters * NACC * 2048.0;
	unsigned long long clk[4];
	(void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
	printf("f64 16x16x4  : %s, %2d acc: %.2f ms  %.1f TFLOP/s   wave: %.1f cycles/MFMA, clock %.0f MHz\n", tag, NACC, ms, fl / ms / 1e9,
		   (double)clk[0] / ((double)iters * NACC), 100.0 * (double)clk[0] / (double)clk[1]);
}

int main() {
	void* buf;
	hipMalloc(&buf, 1 << 24);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int iters = 20000;
	runmix(buf, 10000, 0, e0, e1);
	runmix(buf, 0, 160000, e0, e1);
	runmix(buf, 10000, 160000, e0, e1);
	runmix(buf, 10000, 80000, e0, e1);
	runmix(buf, 10000, 240000, e0, e1);
	runv<16>(buf, 256, iters * 8, e0, e1, "1 WG/CU");
	runv<16>(buf, 512, iters * 8, e0, e1, "2 WG/CU");
	runv<16>(buf, 1024, iters * 8, e0, e1, "4 WG/CU");
	runv<16>(buf, 2048, iters * 8, e0, e1, "8 WG/CU");
	run64<4>(buf, 256 * 3, iters, e0, e1, "3 WG/CU");
	run64<4>(buf, 256 * 4, iters, e0, e1, "4 WG/CU");
	run64<8>(buf, 256 * 4, iters, e0, e1, "4 WG/CU");
	run64<4>(buf, 256 * 6, iters, e0, e1, "6 WG/CU");
	run64<4>(buf, 256 * 8, iters, e0, e1, "8 WG/CU");
	run64<2>(buf, 256 * 8, iters, e0, e1, "8 WG/CU");
	run64<1>(buf, 256 * 8, iters, e0, e1, "8 WG/CU");
	run64<8>(buf, 256, iters, e0, e1, "1 WG/CU");
	run64<16>(buf, 256, iters / 2, e0, e1, "1 WG/CU");
	run64<8>(buf, 512, iters, e0, e1, "2 WG/CU");
	run64<16>(buf, 512, iters / 2, e0, e1, "2 WG/CU");
	run64<16>(buf, 512, iters * 4, e0, e1, "2 WG/CU long");
	for (int wpb = 1; wpb <= 2; wpb++) {  // workgroups per CU
		int grid = 256 * wpb;
		float ms;
		for (int rep = 0; rep < 3; rep++) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(k_f64<4>, dim3(grid), dim3(256), 0, 0, (double*)buf, iters);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			hipEventElapsedTime(&ms, e0, e1);
		}
		double fl = (double)grid * 4 * iters * 4 * 2048.0;
		printf("f64 16x16x4  : %d WG/CU x 4 waves, 4 acc: %.2f ms  %.1f TFLOP/s\n", wpb, ms, fl / ms / 1e9);
		for (int rep = 0; rep < 3; rep++) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, (float*)buf, iters);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			hipEventElapsedTime(&ms, e0, e1);
		}
		fl = (double)grid * 4 * iters * 4 * 4096.0;
		printf("f32 32x32x2  : %d WG/CU x 4 waves, 4 acc: %.2f ms  %.1f TFLOP/s\n", wpb, ms, fl / ms / 1e9);
	}
	return 0;
}
