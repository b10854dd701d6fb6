// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand/result lane maps (by indicator inputs) and issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {  // out[la*64+lb] = bitmask-free: first D lane that is nonzero (or -1), count in high bits
	const int lane = threadIdx.x;
	for (int la = 0; la < 64; la++)
		for (int lb = 0; lb < 64; lb++) {
			double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
			double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
			unsigned long long m = __ballot(d != 0.0);
			if (lane == 0) out[la * 64 + lb] = m ? (__ffsll((long long)m) - 1) | (__popcll(m) << 8) : -1;
		}
}
__global__ void __launch_bounds__(256) rate(double* out, int iters) {
	double acc[8];
	for (int i = 0; i < 8; i++) acc[i] = 0.0;
	double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
	}
	double s = 0;
	for (int i = 0; i < 8; i++) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
	int* d;
	hipMalloc(&d, 4096 * sizeof(int));
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
	std::vector<int> h(4096);
	hipMemcpy(h.data(), d, 4096 * sizeof(int), hipMemcpyDeviceToHost);
	// print for each la the list of lb that produce output and the D lane
	for (int la = 0; la < 64; la += 1) {
		printf("A lane %2d:", la);
		for (int lb = 0; lb < 64; lb++)
			if (h[la * 64 + lb] >= 0) printf(" (B%d->D%d x%d)", lb, h[la * 64 + lb] & 255, h[la * 64 + lb] >> 8);
		printf("\n");
	}
	double* o;
	hipMalloc(&o, 512 * 256 * 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int iters = 20000;
	hipLaunchKernelGGL(rate, dim3(512), dim3(256), 0, 0, o, 100);
	hipEventRecord(e0);
	hipLaunchKernelGGL(rate, dim3(512), dim3(256), 0, 0, o, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	double flops = 512.0 * 4 * iters * 8 * (4 * 4 * 4 * 4 * 2);
	printf("4x4x4 f64: %.3f ms, %.1f TFLOP/s\n", ms, flops / ms / 1e9);
	return 0;
}
