// How fast do K1's digit-plane stores go?  6 planes of (rows / 32) x nks images of 1 KB ([row][32 bytes]); a workgroup of 256 threads
// owns 4 rows and writes, per step of 1024 cells, for each of its rows and each plane one 4-byte word per lane: 8 pieces of 32 bytes,
// 1 KB apart, per wave instruction (pattern A).  Against it: the same bytes with the four rows' words gathered into one 16-byte store
// per lane (B: 8 pieces of 128 bytes per instruction), and a workgroup that owns a whole 32-row block and stores 16 bytes per lane
// contiguously (C: one full 1 KB image per wave instruction).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/store_probe tools/store_probe.hip && ./tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(256) k_store(char* __restrict__ q, int64_t plane_bytes, int64_t nks, int64_t rows) {
	const int tid = threadIdx.x;
	if (MODE == 2) {  // a workgroup per 32-row block: lane l of wave w writes bytes 16 l .. 16 l + 15 of image 4 i + w
		const int64_t ib = blockIdx.x;
		for (int64_t ks = tid >> 6; ks < nks; ks += 4)
#pragma unroll
			for (int s = 0; s < 6; s++) {
				char* dst = q + s * plane_bytes + (ib * nks + ks) * 1024 + (tid & 63) * 16;
				*reinterpret_cast<uint4*>(dst) = make_uint4(tid, s, (unsigned)ks, 7u);
			}
		return;
	}
	const int64_t row0 = (int64_t)blockIdx.x * 4;
	for (int64_t k = (int64_t)tid * 4; k < nks * 32; k += 1024) {
		const int64_t ks = k >> 5;
		const int kk = (int)(k & 31);
		if (MODE == 0) {
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int64_t row = row0 + r;
				char* dst = q + ((row >> 5) * nks + ks) * 1024 + (row & 31) * 32 + kk;
#pragma unroll
				for (int s = 0; s < 6; s++) *reinterpret_cast<unsigned*>(dst + s * plane_bytes) = (unsigned)(tid + s + r);
			}
		} else {  // (not K1's layout: 4 rows x 4 cells side by side -- only to time 16-byte pieces)
			char* dst = q + ((row0 >> 5) * nks + ks) * 1024 + (row0 & 31) * 32 + kk * 4;
#pragma unroll
			for (int s = 0; s < 6; s++) *reinterpret_cast<uint4*>(dst + s * plane_bytes) = make_uint4(tid, s, 1u, 2u);
		}
	}
}

int main() {
	const int64_t rows = 3840, n = 500000, nks = (n + 31) / 32, plane = (rows / 32) * nks * 1024;
	char* q;
	if (hipMalloc(&q, 6 * plane) != hipSuccess) return 1;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	auto time = [&](auto kern, int grid, const char* what) {
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, q, plane, nks, rows);
		(void)hipEventRecord(e0);
		for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, q, plane, nks, rows);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		float ms = 0;
		(void)hipEventElapsedTime(&ms, e0, e1);
		ms /= 3;
		printf("%-72s %.3f ms  (%.1f GB: %.2f TB/s)\n", what, ms, 6 * plane / 1e9, 6 * plane / ms / 1e9);
	};
	time(k_store<0>, (int)(rows / 4), "A: K1's stores (4 B per lane, 32-byte pieces 1 KB apart)");
	time(k_store<1>, (int)(rows / 4), "B: 16 B per lane, 128-byte pieces 1 KB apart");
	time(k_store<2>, (int)(rows / 32), "C: a workgroup per 32-row block, whole 1 KB images (120 workgroups)");
	return 0;
}
