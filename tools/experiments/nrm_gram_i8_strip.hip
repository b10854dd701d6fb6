// K2 (integer engine), the ragged edge of a symmetric problem as a kernel of its own (round 6).
// NOT PART OF THE LIBRARY: built, exact (tools/experiments/strip_parity.py) and measured not to pay -- 2.136 ms against 2.120 in one launch on configs[1]
// (profiles/r06_k2_edge_strip.txt): the strip reads all 302 MB of digit planes once for a quarter tile column of matrix-core work.  Kept for the record.
//
// 5000 genes = 39 x 128 + 8: the 39 off-diagonal tiles of the last tile column hold 8 valid columns each and cost k_gram_i8 a whole tile's time (two
// compute waves on ONE SIMD, the stage's 48 KB of DMA and the barrier per k-step whatever is computed): 4992 genes 2.061 ms, 5000 genes 2.150 ms on one
// box.  Two treatments INSIDE that kernel's geometry were measured and lost (a launch of its own through the same kernel: 0.136 ms for the 0.09 it saves;
// four waves per edge tile: slower -- nrm_gram_i8.hip, profiles/r06_k2_edge_*.txt).  This is the other shape DESIGN priced: the strip
//     dot[i, e0 + j],  i = every row,  j < 32 (the rows of the LAST 32-row block that holds valid rows, e0 = its first row, a multiple of 128),
// is 1/4 of a tile column's matrix-core work and needs no LDS at all: a wave takes one 32-row block i and a range of k-steps, reads the two operands'
// 1 KB digit images straight from global memory into registers (the image IS the MFMA operand: lane l holds 16 bytes of row l & 31 -- nrm_gram_i8.hip's
// layout note), runs the same 21 digit-pair MFMAs per k-step into the same 6 accumulator sets, and hands its piece over as the two EXACT 64-bit integers
// of k_gram_i8's flush; k_gram_i8_strip_sum adds the pieces of a block as integers and rounds ONCE -- every strip entry is the correctly rounded exact
// integer whatever the split (k_gram_i8's split tiles add rounded pieces), so the numbers K3's correction and guard see are the engine's own.
// gram_i8_impl (nrm_gram_i8.hip) plans the symmetric launch without the last tile row / column and calls this for the rest.
#include "nrm_common.h"
#include "nrm_host_logic.h"

typedef int i4s_t __attribute__((ext_vector_type(4)));
typedef int i16s_t __attribute__((ext_vector_type(16)));

#define QS_K 32       // cells per k-step (nrm_gram_i8.hip: QK)
#define QS_CHUNK 512  // k-steps an int32 accumulator set can take (nrm_gram_i8.hip: QCHUNK)

// one piece: block ib = blockIdx.x against the edge block, k-steps of split blockIdx.y
template <int NS>
__global__ void __launch_bounds__(64) k_gram_i8_strip(const char* __restrict__ Q, int64_t plane, int64_t nks, int eb_blk, int ksplit,
													  long long* __restrict__ part) {
	const int lane = threadIdx.x;
	const int ib = blockIdx.x, y = blockIdx.y;
	const int k0 = (int)((int64_t)nks * y / ksplit), k1 = (int)((int64_t)nks * (y + 1) / ksplit);
	const int r = lane & 31;
	const int pos = (2 * r + ((lane >> 5) ^ ((r >> 3) & 1))) * 16;  // (the operand read of nrm_gram_i8.hip, from the image in global memory)
	const char* pa = Q + ((int64_t)ib * nks) * 1024 + pos;
	const char* pe = Q + ((int64_t)eb_blk * nks) * 1024 + pos;
	i16s_t acc[NS];
#pragma unroll
	for (int w = 0; w < NS; w++)
#pragma unroll
		for (int j = 0; j < 16; j++) acc[w][j] = 0;
	i4s_t fa[2][NS], fe[2][NS];
	auto load = [&](int buf, int ks) {
#pragma unroll
		for (int s = 0; s < NS; s++) {
			fa[buf][s] = *reinterpret_cast<const i4s_t*>(pa + s * plane + (int64_t)ks * 1024);
			fe[buf][s] = *reinterpret_cast<const i4s_t*>(pe + s * plane + (int64_t)ks * 1024);
		}
	};
	auto step = [&](int buf) {
#pragma unroll
		for (int s = 0; s < NS; s++)
#pragma unroll
			for (int t = NS - 1 - s; t < NS; t++) acc[s + t - (NS - 1)] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[buf][s], fe[buf][t], acc[s + t - (NS - 1)], 0, 0, 0);
	};
	if (k0 < k1) {
		load(0, k0);
		int ks = k0;
		for (; ks + 2 <= k1 - 1; ks += 2) {  // two k-steps per trip: the buffers keep their names, the next images are in flight while these are contracted
			load(1, ks + 1);
			step(0);
			load(0, ks + 2);
			step(1);
		}
		if (ks + 1 <= k1 - 1) {
			load(1, ks + 1);
			step(0);
			step(1);
		} else {
			step(0);
		}
	}
	// sum_w acc_w 256^w as two exact 64-bit integers (weights 0..2 and 3..NS-1), as k_gram_i8's flush forms them
	long long* o = part + (((int64_t)ib * ksplit + y) * 16) * 128 + lane;
#pragma unroll
	for (int q = 0; q < 16; q++) {
		long long lo = 0, hi = 0;
#pragma unroll
		for (int w = 0; w < NS; w++) {
			if (w < 3)
				lo += (long long)acc[w][q] << (8 * w);
			else
				hi += (long long)acc[w][q] << (8 * (w - 3));
		}
		o[q * 128] = lo;
		o[q * 128 + 64] = hi;
	}
}

// the pieces of block blockIdx.x added as integers, rounded once, scaled, stored: dot[ib * 32 + rr, e0 + (lane & 31)] for the edge's valid columns
template <int NS>
__global__ void __launch_bounds__(64) k_gram_i8_strip_sum(const long long* __restrict__ part, int ksplit, const int* __restrict__ ex, double* __restrict__ C,
														  int64_t ldc, int rows, int e0) {
	const int lane = threadIdx.x, ib = blockIdx.x;
	const int col = e0 + (lane & 31);
	const bool col_ok = col < rows;
	const int ecol = col_ok ? ex[col] : 0;
#pragma unroll
	for (int q = 0; q < 16; q++) {
		long long lo = 0, hi = 0;
		for (int y = 0; y < ksplit; y++) {
			const long long* p = part + (((int64_t)ib * ksplit + y) * 16 + q) * 128 + lane;
			lo += p[0];
			hi += p[64];
		}
		const int rr = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);  // (the accumulator layout of v_mfma_i32_32x32x32_i8: nrm_gram_i8.hip's flush)
		const int row = ib * 32 + rr;
		if (col_ok && row < rows) {
			double v = fma((double)hi, 16777216.0, (double)lo);
			v = ldexp(v, ex[row] + ecol + 8 * (NS - 1));
			C[(int64_t)row * ldc + col] = v;
		}
	}
}

// Bytes of workspace the strip of a problem needs (0: no strip for this shape).  rows: valid rows; the strip exists when the last 128-row tile holds
// 1..32 of them and there is a tile before it.
extern "C" int64_t nrm_gram_i8_strip_bytes(int64_t rows, int64_t k_pad) {
	if (rows <= 128 || rows % 128 == 0 || rows % 128 > 32) return 0;
	const int64_t nks = (k_pad + QS_K - 1) / QS_K;
	const int64_t nblk = (rows + 31) / 32;
	int64_t ksplit = (nks + QS_CHUNK - 1) / QS_CHUNK;
	if (ksplit < 8) ksplit = nks < 8 ? nks : 8;
	return nblk * ksplit * 16 * 128 * (int64_t)sizeof(long long);
}

// d_q / d_ex: the symmetric operand's digit planes (dense, m_pad rows) and exponents; the strip's entries are WRITTEN (not added) into d_dot.
extern "C" int nrm_gram_i8_strip(const void* d_q, const int32_t* d_ex, int64_t plane_bytes, int64_t m_pad, int64_t k_pad, int nslices, double* d_dot,
								 int64_t ldd, int64_t rows, void* d_work, int64_t work_bytes, void* stream) {
	NRM_REQUIRE(nslices == 5 || nslices == 6, "nrm_gram_i8_strip: 5 or 6 slices");
	const int64_t need = nrm_gram_i8_strip_bytes(rows, k_pad);
	NRM_REQUIRE(need > 0 && rows <= m_pad && m_pad % 128 == 0 && m_pad - rows < 128, "nrm_gram_i8_strip: no ragged edge of at most 32 rows in this shape");
	NRM_REQUIRE(d_q && d_ex && d_dot && d_work && work_bytes >= need && ldd >= m_pad, "nrm_gram_i8_strip: null pointer, workspace or pitch too small");
	const int64_t nks = (k_pad + QS_K - 1) / QS_K;
	const int64_t plane = plane_bytes ? plane_bytes : (m_pad / 32) * nks * 1024;
	const int e0 = (int)(rows / 128 * 128);
	const int nblk = (int)((rows + 31) / 32);
	int64_t ksplit = (nks + QS_CHUNK - 1) / QS_CHUNK;
	if (ksplit < 8) ksplit = nks < 8 ? nks : 8;
	NRM_REQUIRE(ksplit < 65536 && rows < (1 << 30), "nrm_gram_i8_strip: problem too large");
	hipStream_t st = (hipStream_t)stream;
	if (nslices == 5) {
		hipLaunchKernelGGL(k_gram_i8_strip<5>, dim3((unsigned)nblk, (unsigned)ksplit), dim3(64), 0, st, (const char*)d_q, plane, nks, e0 / 32, (int)ksplit, (long long*)d_work);
		hipLaunchKernelGGL(k_gram_i8_strip_sum<5>, dim3((unsigned)nblk), dim3(64), 0, st, (const long long*)d_work, (int)ksplit, d_ex, d_dot, ldd, (int)rows, e0);
	} else {
		hipLaunchKernelGGL(k_gram_i8_strip<6>, dim3((unsigned)nblk, (unsigned)ksplit), dim3(64), 0, st, (const char*)d_q, plane, nks, e0 / 32, (int)ksplit, (long long*)d_work);
		hipLaunchKernelGGL(k_gram_i8_strip_sum<6>, dim3((unsigned)nblk), dim3(64), 0, st, (const long long*)d_work, (int)ksplit, d_ex, d_dot, ldd, (int)rows, e0);
	}
	return nrm_check_launch("k_gram_i8_strip");
}
