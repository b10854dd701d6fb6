// Microbenchmark: sustained rate of v_mfma_i32_16x16x64_i8 / 32x32x32_i8 on this GPU with pseudo-random operands
// (register operands only).  Puts a number on DESIGN.md's backlog item "exact integer engine for K2":
// a fixed-point Gram needs ~15-21 int8 slice products per fp64 product.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_i8_peak tools/mfma_i8_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k_i8_16(int* out, int iters) {
	i4 acc[NACC];
	for (int i = 0; i < NACC; i++) acc[i] = (i4){0, 0, 0, 0};
	unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	i4 a, b;
	for (int j = 0; j < 4; j++) {
		s = s * 1664525u + 1013904223u;
		a[j] = (int)(s & 0x3f3f3f3f);
		s = s * 1664525u + 1013904223u;
		b[j] = (int)(s & 0x3f3f3f3f);
	}
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
	}
	int r = 0;
	for (int i = 0; i < NACC; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_i8_32(int* out, int iters) {
	i16 acc[NACC];
	for (int i = 0; i < NACC; i++)
		for (int j = 0; j < 16; j++) acc[i][j] = 0;
	unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 999u;
	i4 a, b;
	for (int j = 0; j < 4; j++) {
		s = s * 1664525u + 1013904223u;
		a[j] = (int)(s & 0x3f3f3f3f);
		s = s * 1664525u + 1013904223u;
		b[j] = (int)(s & 0x3f3f3f3f);
	}
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
	}
	int r = 0;
	for (int i = 0; i < NACC; i++)
		for (int j = 0; j < 16; j++) r += acc[i][j];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
	int* buf;
	if (hipMalloc(&buf, 1 << 24) != hipSuccess) return 1;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	const int iters = 40000;
	for (int wg = 1; wg <= 2; wg++) {
		const int grid = 256 * wg;
		float ms = 0;
		for (int rep = 0; rep < 3; rep++) {
			(void)hipEventRecord(e0);
			hipLaunchKernelGGL(k_i8_16<8>, dim3(grid), dim3(256), 0, 0, buf, iters);
			(void)hipEventRecord(e1);
			(void)hipEventSynchronize(e1);
			(void)hipEventElapsedTime(&ms, e0, e1);
		}
		printf("i8 16x16x64: %d WG/CU x 4 waves, 8 acc: %.2f ms  %.0f TOP/s\n", wg, ms, (double)grid * 4 * iters * 8 * (2.0 * 16 * 16 * 64) / ms / 1e9);
		for (int rep = 0; rep < 3; rep++) {
			(void)hipEventRecord(e0);
			hipLaunchKernelGGL(k_i8_32<4>, dim3(grid), dim3(256), 0, 0, buf, iters);
			(void)hipEventRecord(e1);
			(void)hipEventSynchronize(e1);
			(void)hipEventElapsedTime(&ms, e0, e1);
		}
		printf("i8 32x32x32: %d WG/CU x 4 waves, 4 acc: %.2f ms  %.0f TOP/s\n", wg, ms, (double)grid * 4 * iters * 4 * (2.0 * 32 * 32 * 32) / ms / 1e9);
	}
	return 0;
}
