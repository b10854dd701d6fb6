// Feasibility probe for the exact int8-sliced Gram kernel: the inner loop only (operands resident in LDS, no global loads).
// A wave owns a 64x64 output tile = 2x2 MFMA tiles of v_mfma_i32_32x32x32_i8; NS radix-256 slices per operand; slice pairs
// (s,t) with s + t >= NS - 1 are accumulated into one i32 accumulator set per weight w = s + t (NS sets: 4 * 16 * NS VGPRs).
//   hipcc --offload-arch=gfx950 -O3 -o i8gram_probe tools/i8gram_probe.hip && ./i8gram_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));

template <int NS, bool BARRIER>
__global__ void __launch_bounds__(256) k_probe(int* out, int ksteps) {
	__shared__ __attribute__((aligned(16))) char lds[2][NS][128 * 32];  // [A|B][slice][128 rows x 32 bytes]
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int wm = wid >> 1, wn = wid & 1;
	for (int i = tid; i < (int)sizeof(lds) / 4; i += 256) ((int*)lds)[i] = (int)((i * 2654435761u) & 0x3f3f3f3fu);
	__syncthreads();
	i16 acc[NS][4];
	for (int w = 0; w < NS; w++)
		for (int q = 0; q < 4; q++)
			for (int j = 0; j < 16; j++) acc[w][q][j] = 0;
	const int r = lane & 31, h = lane >> 5;
	const int pos = (2 * r + (h ^ ((r >> 3) & 1))) * 16;
	for (int ks = 0; ks < ksteps; ks++) {
		i4 fa[NS][2], fb[NS][2];
#pragma unroll
		for (int s = 0; s < NS; s++)
#pragma unroll
			for (int i = 0; i < 2; i++) {
				fa[s][i] = *reinterpret_cast<const i4*>(&lds[0][s][(wm * 64 + i * 32) * 32 + pos]);
				fb[s][i] = *reinterpret_cast<const i4*>(&lds[1][s][(wn * 64 + i * 32) * 32 + pos]);
			}
#pragma unroll
		for (int s = 0; s < NS; s++)
#pragma unroll
			for (int t = 0; t < NS; t++)
				if (s + t >= NS - 1) {
					const int w = s + t - (NS - 1);
#pragma unroll
					for (int i = 0; i < 2; i++)
#pragma unroll
						for (int j = 0; j < 2; j++)
							acc[w][i * 2 + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s][i], fb[t][j], acc[w][i * 2 + j], 0, 0, 0);
				}
		if (BARRIER) __syncthreads();
	}
	int sum = 0;
	for (int w = 0; w < NS; w++)
		for (int q = 0; q < 4; q++)
			for (int j = 0; j < 16; j++) sum += acc[w][q][j];
	out[blockIdx.x * 256 + tid] = sum;
}

// 8 waves per workgroup (2 per SIMD), each 64 rows x 32 columns (2 x 1 MFMA tiles): 2 * 16 * NS accumulator registers
template <int NS, bool BARRIER>
__global__ void __launch_bounds__(512) k_probe8(int* out, int ksteps) {
	__shared__ __attribute__((aligned(16))) char lds[2][NS][128 * 32];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int wm = wid >> 2, wn = wid & 3;
	for (int i = tid; i < (int)sizeof(lds) / 4; i += 512) ((int*)lds)[i] = (int)((i * 2654435761u) & 0x3f3f3f3fu);
	__syncthreads();
	i16 acc[NS][2];
	for (int w = 0; w < NS; w++)
		for (int q = 0; q < 2; q++)
			for (int j = 0; j < 16; j++) acc[w][q][j] = 0;
	const int r = lane & 31, h = lane >> 5;
	const int pos = (2 * r + (h ^ ((r >> 3) & 1))) * 16;
	for (int ks = 0; ks < ksteps; ks++) {
		i4 fa[NS][2], fb[NS];
#pragma unroll
		for (int s = 0; s < NS; s++) {
#pragma unroll
			for (int i = 0; i < 2; i++) fa[s][i] = *reinterpret_cast<const i4*>(&lds[0][s][(wm * 64 + i * 32) * 32 + pos]);
			fb[s] = *reinterpret_cast<const i4*>(&lds[1][s][(wn * 32) * 32 + pos]);
		}
#pragma unroll
		for (int s = 0; s < NS; s++)
#pragma unroll
			for (int t = 0; t < NS; t++)
				if (s + t >= NS - 1) {
					const int w = s + t - (NS - 1);
#pragma unroll
					for (int i = 0; i < 2; i++) acc[w][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s][i], fb[t], acc[w][i], 0, 0, 0);
				}
		if (BARRIER) __syncthreads();
	}
	int sum = 0;
	for (int w = 0; w < NS; w++)
		for (int q = 0; q < 2; q++)
			for (int j = 0; j < 16; j++) sum += acc[w][q][j];
	out[blockIdx.x * 512 + tid] = sum;
}

template <int NS, bool BARRIER>
static void run8(int* buf, const char* name) {
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	const int ksteps = 2000, grid = 256;
	hipLaunchKernelGGL((k_probe8<NS, BARRIER>), dim3(grid), dim3(512), 0, 0, buf, 10);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((k_probe8<NS, BARRIER>), dim3(grid), dim3(512), 0, 0, buf, ksteps);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const int pairs = NS * (NS + 1) / 2;
	const double ops = (double)grid * 8 * ksteps * pairs * 2 * 2.0 * 32 * 32 * 32;
	printf("8 waves %s NS=%d pairs=%d: %.3f ms  %.2f POP/s  = %.1f fp64-equivalent TFLOP/s\n", name, NS, pairs, ms, ops / ms / 1e12, ops / pairs / ms / 1e9);
}

template <int NS, bool BARRIER>
static void run(int* buf, const char* name) {
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	const int ksteps = 2000, grid = 256;
	hipLaunchKernelGGL((k_probe<NS, BARRIER>), dim3(grid), dim3(256), 0, 0, buf, 10);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((k_probe<NS, BARRIER>), dim3(grid), dim3(256), 0, 0, buf, ksteps);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const int pairs = NS * (NS + 1) / 2;
	const double ops = (double)grid * 4 * ksteps * pairs * 4 * 2.0 * 32 * 32 * 32;
	printf("%s NS=%d pairs=%d: %.3f ms  %.2f POP/s  = %.1f fp64-equivalent TFLOP/s (per pair set)\n", name, NS, pairs, ms, ops / ms / 1e12,
		   ops / pairs / ms / 1e9);
}

int main() {
	int* buf;
	if (hipMalloc(&buf, 1 << 22) != hipSuccess) return 1;
	run<5, false>(buf, "no barrier");
	run<5, true>(buf, "barrier   ");
	run<6, false>(buf, "no barrier");
	run<4, false>(buf, "no barrier");
	run8<5, false>(buf, "no barrier");
	run8<5, true>(buf, "barrier   ");
	run8<6, true>(buf, "barrier   ");
	return 0;
}
