// K1's traffic as two kernels over bands of rows (round-4 verdict, item 6): sweep A reads a band (products, maximum, |x|^2), sweep B reads it again and
// writes 0.75 of its size (the digit planes of fp64 rows).  If the band is small enough to sit in the 256 MB Infinity Cache between the two, B's reads
// do not reach HBM -- but is the PAIR any faster than the same two sweeps over the whole matrix, where every byte comes from HBM twice?  Time decides:
// FETCH_SIZE cannot (the fabric-side counters count Infinity-Cache hits too, MI355X_MICROARCH.md).  hipcc --offload-arch=gfx950 -O3 tools/mall_band_probe.hip -o tools/exp/mall_band_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_a(const uint4* __restrict__ p, size_t n16, double* __restrict__ out) {
	double acc = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		const uint4 v = p[i];
		acc += (double)(v.x ^ v.w) * 1e-9 + (double)(v.y ^ v.z) * 1e-9;
	}
	if (acc == 0.123456789) out[0] = acc;
}
// reads 16 B per thread-step, writes 12 B (0.75): three of every four threads store a 16-byte piece
__global__ void __launch_bounds__(256) k_b(const uint4* __restrict__ p, size_t n16, uint4* __restrict__ q) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		const uint4 v = p[i];
		if ((i & 3) != 3) q[i - (i >> 2)] = make_uint4(v.x + 1, v.y, v.z, v.w);
	}
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
	const size_t MB = 1 << 20, TOTAL = 4096 * MB;
	uint4 *x, *q, *big;
	double* out;
	CK(hipMalloc(&x, TOTAL));
	CK(hipMalloc(&q, TOTAL));
	CK(hipMalloc(&big, 1024 * MB));
	CK(hipMalloc(&out, 64));

	for (size_t o = 0; o < TOTAL; o += 1024 * MB) hipMemset((char*)x + o, 1, 1024 * MB);  // (pieces below 4 GiB: one 2^32-byte memset faulted)
	hipMemset(big, 2, 1024 * MB);
	if (hipDeviceSynchronize() != hipSuccess || !x || !q || !big) {
		printf("setup failed\n");
		return 1;
	}
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	auto flush = [&]() { k_a<<<2048, 256>>>(big, 1024 * MB / 16, out); hipDeviceSynchronize(); };
	auto run = [&](size_t band_mb) -> float {  // 0: the two sweeps over the whole matrix
		const size_t band = band_mb ? band_mb * MB : TOTAL;
		flush();
		hipEventRecord(e0, 0);
		for (size_t off = 0; off < TOTAL; off += band) {
			const size_t n16 = (band < TOTAL - off ? band : TOTAL - off) / 16;  // (96 and 48 MB do not divide 4 GiB)
			const uint4* src = x + off / 16;
			uint4* dst = q + (off / 16) / 4 * 3;
			hipLaunchKernelGGL(k_a, dim3(2048), dim3(256), 0, 0, src, n16, out);
			hipLaunchKernelGGL(k_b, dim3(2048), dim3(256), 0, 0, src, n16, dst);
		}
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		if (hipGetLastError() != hipSuccess) printf("error at band %zu\n", band_mb);
		return ms;
	};
	printf("4 GiB of rows: sweep A (read) + sweep B (read again, write 0.75); algorithmic bytes = 1.75 x 4 GiB, moved = 2.75 x\n");
	printf("band MB | launches | ms (best of 4) | algorithmic TB/s | moved TB/s\n");
	for (size_t mb : {(size_t)0, (size_t)512, (size_t)256, (size_t)128, (size_t)96, (size_t)64, (size_t)48, (size_t)32}) {
		float best = 1e9;
		for (int rep = 0; rep < 4; rep++) best = fminf(best, run(mb));
		const double alg = 1.75 * TOTAL / 1e12, mov = 2.75 * TOTAL / 1e12;
		printf("%7zu | %8zu | %8.3f | %8.2f | %8.2f\n", mb, mb ? 2 * (TOTAL / (mb * MB)) : (size_t)2, best, alg / (best * 1e-3), mov / (best * 1e-3));
	}
	// the sweeps alone, whole matrix
	flush();
	hipEventRecord(e0);
	k_a<<<2048, 256>>>(x, TOTAL / 16, out);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	printf("sweep A alone over 4 GiB: %.3f ms = %.2f TB/s\n", ms, TOTAL / 1e12 / (ms * 1e-3));
	hipEventRecord(e0);
	k_b<<<2048, 256>>>(x, TOTAL / 16, q);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	hipEventElapsedTime(&ms, e0, e1);
	printf("sweep B alone over 4 GiB: %.3f ms = %.2f TB/s moved\n", ms, 1.75 * TOTAL / 1e12 / (ms * 1e-3));
	return 0;
}
