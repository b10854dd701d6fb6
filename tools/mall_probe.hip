// Does a band of rows read by one kernel still sit in the 256 MB Infinity Cache (MALL) when the next kernel reads it again, with a
// stream of writes in between?  Decides whether K1 split into a products kernel and a digits kernel over MALL-sized bands of rows can
// replace the second HBM read of every row.  hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o tools/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ p, size_t n16, unsigned* __restrict__ out) {
	unsigned acc = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		const uint4 v = p[i];
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(uint4* __restrict__ p, size_t n16, unsigned s) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(s, s + 1, s + 2, (unsigned)i);
}

int main() {
	const size_t MB = 1 << 20;
	uint4 *a, *b, *big;
	unsigned* out;
	hipMalloc(&a, 512 * MB);
	hipMalloc(&b, 1536 * MB);
	hipMalloc(&big, 2048 * MB);
	hipMalloc(&out, 64);
	hipMemset(a, 1, 512 * MB);
	hipMemset(big, 2, 2048 * MB);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	auto timed = [&](auto fn) {
		hipEventRecord(e0);
		fn();
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		return ms;
	};
	auto flush = [&]() { k_read<<<2048, 256>>>(big, 2048 * MB / 16, out); hipDeviceSynchronize(); };
	printf("band MB | cold read GB/s | re-read at once | re-read after writing 1.5x the band | after writing 3x\n");
	for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384}) {
		const size_t n16 = mb * MB / 16;
		float cold = 0, warm = 0, w15 = 0, w3 = 0;
		for (int rep = 0; rep < 3; rep++) {
			flush();
			cold = timed([&] { k_read<<<2048, 256>>>(a, n16, out); });
			warm = timed([&] { k_read<<<2048, 256>>>(a, n16, out); });
			flush();
			k_read<<<2048, 256>>>(a, n16, out);
			k_write<<<2048, 256>>>(b, n16 * 3 / 2, rep);
			w15 = timed([&] { k_read<<<2048, 256>>>(a, n16, out); });
			flush();
			k_read<<<2048, 256>>>(a, n16, out);
			k_write<<<2048, 256>>>(b, n16 * 3, rep);
			w3 = timed([&] { k_read<<<2048, 256>>>(a, n16, out); });
		}
		const double gb = mb * MB / 1e9;
		printf("%7zu | %8.0f | %8.0f | %8.0f | %8.0f\n", mb, gb / (cold * 1e-3), gb / (warm * 1e-3), gb / (w15 * 1e-3), gb / (w3 * 1e-3));
	}
	return 0;
}
