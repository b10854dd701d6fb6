// Issue rate of the instructions the sparse-design gathers are made of (gfx950): v_add_f64, v_cvt_f64_f32, v_fma_f64, and the integer
// sequence that widens an fp32 to an fp64 without the conversion unit.  hipcc --offload-arch=gfx950 -O3 -o tools/cvt_rate_probe tools/cvt_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, double* __restrict__ out, int iters) {
	float x[8];
	double s[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		x[i] = in[threadIdx.x + 256 * i];
		s[i] = 0.0;
	}
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			if (MODE == 0) {  // add only
				s[i] += 1.25;
			} else if (MODE == 1) {  // cvt + add
				double d;
				asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x[i]));
				s[i] += d;
			} else if (MODE == 2) {  // cvt only (result xor-ed in cheaply is not possible: keep the add out by accumulating every 8th)
				double d;
				asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x[i]));
				if (it == iters - 1) s[i] += d;
			} else if (MODE == 3) {  // integer widening + add
				const unsigned b = __float_as_uint(x[i]);
				unsigned hi, lo;
				hi = ((b >> 3) & 0x0fffffffu) | (b & 0x80000000u);  // (v_lshrrev, v_and_or / v_bfi)
				hi += 0x38000000u;
				lo = b << 29;
				asm volatile("" : "+v"(hi), "+v"(lo));
				s[i] += __hiloint2double((int)hi, (int)lo);
			} else if (MODE == 4) {  // fma f64
				s[i] = fma(s[i], 1.0000001, 1.25);
			} else if (MODE == 5) {  // v_add_f32 for reference
				x[i] += 1.25f;
			}
		}
	}
	double t = 0.0;
#pragma unroll
	for (int i = 0; i < 8; i++) t += s[i] + (double)x[i];
	out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE>
void run(const char* name, const float* in, double* out, int per_iter) {
	const int iters = 20000, blocks = 256 * 8;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, 100);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	// wave-instructions per SIMD: blocks * 4 waves * iters * 8 * per_iter / (256 CUs * 4 SIMDs)
	const double wi = (double)blocks * 4 * iters * 8 * per_iter / 1024.0;
	printf("%-28s %8.3f ms  %6.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / wi, ms * 1e6 / wi * 2.4);
}

int main() {
	float* in;
	double* out;
	hipMalloc(&in, 256 * 8 * 4);
	hipMalloc(&out, 256 * 8 * 256 * 8);
	std::vector<float> h(256 * 8, 1.5f);
	hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	run<0>("v_add_f64", in, out, 1);
	run<4>("v_fma_f64", in, out, 1);
	run<5>("v_add_f32", in, out, 1);
	run<2>("v_cvt_f64_f32", in, out, 1);
	run<1>("v_cvt_f64_f32 + v_add_f64", in, out, 2);
	run<3>("4 int ops + v_add_f64", in, out, 5);
	return 0;
}
