// Ablation for an integer formulation of the streaming de kernel (K2s, csrc/nrm_gram_skinny.hip): how fast can
// the expression rows be turned into fixed-point digit planes ON THE FLY, between the HBM stream and the matrix cores?
// The streaming kernel reads every fp32 expression value once (8 GB on BASELINE configs[2]) and is bound today by the fp64 matrix-core
// work of its 20 Z rows (1.85 ms = 4.3 TB/s).  On the int8 matrix cores the products would cost a third of that -- if each value
// can be scaled by its row's power of two, rounded to 38-bit fixed point (5 digits: what the accuracy guard of csrc/nrm_fix.h can
// certify at 100 000 cells) and cut into digits at the rate the stream delivers it: 2e9 values in ~1.3 ms.
// This probe measures exactly that stage alone, as generously as possible: the rows come from HBM with 16-byte loads, the digits
// go to LDS (5 x ds_write_b32 per 4 values, no bank conflicts), nothing is contracted and nothing is written back.
//   hipcc --offload-arch=gfx950 -O3 -o tools/quant_probe tools/quant_probe.hip && ./tools/quant_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int NS>
__device__ __forceinline__ void digits4(const double (&v)[4], int sh, unsigned (&w)[NS]) {  // as csrc/nrm_digits.h
	unsigned long long bias = 0;
#pragma unroll
	for (int s = 0; s < NS - 1; s++) bias |= 0x80ull << (8 * s);
	unsigned lo[4], hi[4];
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const double t = ldexp(v[i], -sh) + 6755399441055744.0;
		const unsigned long long b = ((unsigned long long)__double_as_longlong(t) + bias) ^ bias;
		lo[i] = (unsigned)b;
		hi[i] = (unsigned)(b >> 32);
	}
#pragma unroll
	for (int s = 0; s < NS; s++) {
		const unsigned* src = s < 4 ? lo : hi;
		const unsigned sel = 0x0c0c0000u | ((4u + (s & 3)) << 8) | (unsigned)(s & 3);
		const unsigned t01 = __builtin_amdgcn_perm(src[1], src[0], sel), t23 = __builtin_amdgcn_perm(src[3], src[2], sel);
		w[s] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
	}
}

// the same for fp32 input with integer arithmetic only: mantissa and exponent by hand, one 64-bit shift per value
template <int NS>
__device__ __forceinline__ void digits4_f32(const float (&v)[4], int sh, unsigned (&w)[NS]) {
	unsigned long long bias = 0;
#pragma unroll
	for (int s = 0; s < NS - 1; s++) bias |= 0x80ull << (8 * s);
	unsigned lo[4], hi[4];
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const unsigned u = __float_as_uint(v[i]);
		const int e = (int)((u >> 23) & 0xff);
		const unsigned long long m = e ? ((u & 0x7fffffu) | 0x800000u) : 0u;  // (subnormals count as 0: 1e-38 of anything measurable)
		const int up = e - 150 - sh;  // value = m 2^(e - 150); q = value 2^-sh
		long long q = up >= 0 ? (long long)(m << (up & 63)) : (long long)((m + (1ull << ((-up - 1) & 63))) >> ((-up) & 63));  // round half up
		if (up < -40) q = 0;
		q = (u >> 31) ? -q : q;
		const unsigned long long b = ((unsigned long long)q + bias) ^ bias;
		lo[i] = (unsigned)b;
		hi[i] = (unsigned)(b >> 32);
	}
#pragma unroll
	for (int s = 0; s < NS; s++) {
		const unsigned* src = s < 4 ? lo : hi;
		const unsigned sel = 0x0c0c0000u | ((4u + (s & 3)) << 8) | (unsigned)(s & 3);
		const unsigned t01 = __builtin_amdgcn_perm(src[1], src[0], sel), t23 = __builtin_amdgcn_perm(src[3], src[2], sel);
		w[s] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
	}
}

template <int NS, int MODE>  // MODE 0: load only; 1: fp64 digit extraction; 2: integer digit extraction
__global__ void __launch_bounds__(256) k_quant(const float* __restrict__ y, int64_t rows, int64_t n, const int* __restrict__ exps, int* __restrict__ sink) {
	__shared__ unsigned planes_[NS][256 * 4];
	volatile unsigned (*planes)[256 * 4] = planes_;  // (volatile: every store is issued, as a consumer wave would need it)
	const int tid = threadIdx.x;
	unsigned acc = 0;
	for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
		const int sh = exps[row];
		const float* x = y + row * n;
		for (int64_t k = (int64_t)tid * 4; k < n; k += 4096) {
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int64_t kk = k + j * 1024;
				if (kk < n) {
					const float4 t = *reinterpret_cast<const float4*>(x + kk);
					unsigned w[NS];
					if (MODE == 1) {
						const double v[4] = {t.x, t.y, t.z, t.w};
						digits4<NS>(v, sh, w);
					} else if (MODE == 2) {
						const float v[4] = {t.x, t.y, t.z, t.w};
						digits4_f32<NS>(v, sh, w);
					} else {
#pragma unroll
						for (int s = 0; s < NS; s++) w[s] = __float_as_uint(t.x) + s;
					}
					if (MODE) {
#pragma unroll
						for (int s = 0; s < NS; s++) planes[s][j * 256 + tid] = w[s];
					} else
						acc += w[0] ^ w[NS - 1];
				}
			}
		}
	}
	__syncthreads();
	if (MODE) acc = planes[0][tid] ^ planes[NS - 1][(tid * 7) & 1023];
	if (acc == 0x12345678u) sink[0] = 1;
}

int main() {
	const int64_t rows = 20000, n = 100000;
	float* y;
	int *e, *sink;
	if (hipMalloc(&y, rows * n * 4) != hipSuccess || hipMalloc(&e, rows * 4) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
	(void)hipMemset(y, 0x3c, rows * n * 4);
	(void)hipMemset(e, 0xff, rows * 4);  // sh = -1
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	auto time = [&](auto kern, const char* what) {
		for (int grid : {1024, 2048, 4096}) {
			hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, y, rows, n, e, sink);
			(void)hipEventRecord(e0);
			for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, y, rows, n, e, sink);
			(void)hipEventRecord(e1);
			(void)hipEventSynchronize(e1);
			float ms = 0;
			(void)hipEventElapsedTime(&ms, e0, e1);
			ms /= 5;
			printf("%-52s grid %4d: %.3f ms = %.2f TB/s of fp32 input, %.0f G values/s\n", what, grid, ms, rows * n * 4 / ms / 1e9, rows * n / ms / 1e6);
		}
	};
	time(k_quant<5, 0>, "stream only (16-byte loads)");
	time(k_quant<5, 1>, "5 digits, fp64 rounding (csrc/nrm_digits.h) -> LDS");
	time(k_quant<5, 2>, "5 digits, integer arithmetic -> LDS");
	time(k_quant<4, 2>, "4 digits, integer arithmetic -> LDS");
	return 0;
}
