// K2 (integer engine): dot[i,j] = sum_k A[i,k] B[j,k] EXACTLY for fixed-point operands, on the int8 matrix cores
// (v_mfma_i32_32x32x32_i8: 3.9 POP/s measured on this chip against 72 TFLOP/s for v_mfma_f64_16x16x4_f64).
// Replaces np.matmul(dy1, dx1.T) at association.py:234 like nrm_gram.hip does; same tile order, schedule and fix-up.
//
// Arithmetic.  Row i of an operand is scaled by a power of two so that |x| <= 2^e_i and rounded ONCE to a fixed-point
// integer q = rint(x 2^(B - e_i)), B = 8 NS - 2 bits (NS = 6 slices: 46 bits, i.e. 1.4e-14 of the row's largest entry --
// the rounding an fp64 dot product of ~1e4 terms carries anyway; NS = 5: 38 bits).  q is cut into NS balanced radix-256
// digits d_s in [-128, 127] (top digit within +-64): q = sum_s d_s 256^s.  Then
//     q_i . q_j = sum_{s,t} 256^(s+t) sum_k d_is[k] d_jt[k]
// and every inner sum is an int8 x int8 -> int32 contraction that the matrix cores compute without any rounding.  Slice
// pairs with s + t < NS - 1 are dropped (each sits near 2^-(8 NS + 2) of full scale); the
// NS (NS + 1) / 2 pairs kept are accumulated into ONE int32 accumulator set per weight w = s + t (NS sets): |d d'| <= 2^14,
// at most NS pairs per set, so 2^14 NS K < 2^31 bounds a chunk to K <= 16 384 cells, after which the sets are combined in
// fp64 (sum_w acc_w 256^w formed as two exact 64-bit integers joined by one fma: the correctly rounded exact value) and added
// to the output.  Chunks and the partial pieces of the
// stream-K schedule are combined in fp64 in a fixed order: bitwise reproducible from run to run like the fp64 kernel.
// Accuracy: the dropped digit products are the error, not the quantisation -- measured |delta r| <= 3e-14 (2e-15 typical at
// 10 000 Gaussian cells; it grows with (max / rms)^2 of the rows and as 1 / sqrt(cells)), P-values to 4e-10 relative; on the
// diagonal (identical rows) the dropped squares add coherently (4e-13), which nobody reads (sums of squares come from K1).
//
// Operand layout (written by k_quantize_rows, read back by plain DMA): plane s holds, for every block of 32 rows and every
// k-step of 32 cells, one 1 KB image [row][32 bytes] whose two 16-byte halves of row r are swapped when (r >> 3) & 1 -- the
// bank-conflict-free order for the MFMA operand reads (lane l reads 16 bytes of row l & 31, half l >> 5; a dot product
// does not care which 16 of the 32 cells an instruction takes first as long as both operands agree).  One DMA instruction
// (global_load_lds_dwordx4, issued from inline asm: see nrm_gram_skinny.hip) moves one such image: contiguous 1 KB.
//
// Geometry: workgroup tile 128 x 128, 8 waves (2 per SIMD) of 64 rows x 32 columns = 2 MFMA tiles, 2 * 16 * NS
// accumulator registers; a stage is one k-step of both operand panels (8 NS KB), ring of 3 stages (144 KB at NS = 6),
// every wave issues NS DMA instructions per stage (the NS slices of one 32-row block), one barrier per stage.
// Inner loop alone, operands resident in LDS (tools/i8gram_probe.hip): 3.5 POP/s = 231 fp64-equivalent TFLOP/s at NS = 5,
// 170 at NS = 6, against 68 executed by the fp64 kernel.  The chip clocks down under int8 MFMA load (tools/clock_probe.sh:
// 1.80 GHz effective in this kernel, 2.34 GHz in the fp64 kernel), so cycles saved come back partly as clock.
// Measured and rejected (C2, 2.15-2.25 ms as is): two 4-wave workgroups per CU on 128 x 64 half tiles with a ring of 2 (2.64 ms;
// removing barrier and vmcnt waits altogether gains only 3 %: lockstep is not the cost); issuing the refill from waves 4-7 half
// way through their MFMAs (4.05 ms, the branch breaks the MFMA block); a memory-clobbering asm between operand reads and MFMAs
// (7.1 ms: every read drains first); s_setprio(1) around the MFMA block (11 ms).
#include <type_traits>
#include "nrm_gram_sched.h"
#include "nrm_digits.h"
#include "nrm_fix.h"

#define QK 32        // cells per k-step (one MFMA)
#define QCHUNK 512   // k-steps per int32 accumulation chunk (16 384 cells)
#define QD 3         // stages in the LDS ring
#ifndef QI_EDGE
#define QI_EDGE 0    // 1: edge tiles on four waves of one sub-tile each (round 6: built, parity-green, measured NOT to help -- see gram_piece_i8; kept for A/B builds)
#endif
#ifndef QI_EXP
#define QI_EXP 0     // timing experiments (tools/build_exp.sh): 1 no DMA, 2 every DMA from k-steps 0-3 (L2 resident), 4 no barrier
#endif

typedef int i4_t __attribute__((ext_vector_type(4)));
typedef int i16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void q_dma16(const void* gsrc, unsigned lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
				 : "=&s"(keep)
				 : "v"(gsrc), "s"(lds_dst)
				 : "memory");
}

// the same without the compiler-level memory clobber, for use between sched_barriers inside the MFMA stream (ordering against
// the LDS reads of later k-steps is carried by the vmcnt wait and the workgroup barrier, both of which clobber memory)
__device__ __forceinline__ void q_dma16_nc(const void* gsrc, unsigned lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
				 : "=&s"(keep)
				 : "v"(gsrc), "s"(lds_dst));
}

// ---- quantiser: fp64 rows -> NS digit planes in the tiled layout + one exponent per row ----------------------------
// One wave per row (4 rows per workgroup, consecutive rows of one 32-row block).  x = q 2^exps[row] + rounding.
template <int NS>
__global__ void __launch_bounds__(256) k_quantize_rows(const double* __restrict__ X, int64_t kx, int64_t ldx, char* __restrict__ Q,
													   int64_t plane_bytes, int64_t nks, int* __restrict__ exps, double* __restrict__ fix, double n_cells) {
	constexpr int B = 8 * NS - 2;
	const int lane = threadIdx.x & 63;
	const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const double* x = X + row * ldx;
	double mx = 0.0, ssq = 0.0;
	for (int64_t k = (int64_t)lane * 2; k < kx; k += 128) {
		const d2_t v = *reinterpret_cast<const d2_t*>(x + k);
		mx = fmax(mx, fmax(fabs(v[0]), fabs(v[1])));
		ssq = fma(v[0], v[0], fma(v[1], v[1], ssq));
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
	int e = 0;
	if (mx > 0.0 && mx < INFINITY) (void)frexp(mx, &e);  // mx = m 2^e, m in [0.5, 1): |x| < 2^e
	const int sh = e - B;
	if (lane == 0) exps[row] = sh;
	const int64_t ib = row >> 5;
	const int r = (int)(row & 31);
	char* qrow = Q + (ib * nks) * 1024 + (2 * r) * 16;
	const int flip = (r >> 3) & 1;
	int dsum[NS - 1];
	unsigned dsq[NS - 1];
	unsigned long long dsq64[NS - 1];
	unsigned steps = 0;
#pragma unroll
	for (int s = 0; s < NS - 1; s++) {
		dsum[s] = 0;
		dsq[s] = 0u;
		dsq64[s] = 0ull;
	}
	for (int64_t k = (int64_t)lane * 4; k < nks * QK; k += 256) {
		double v[4] = {0.0, 0.0, 0.0, 0.0};
		if (k < kx) {  // kx is a multiple of 16: a group of 4 cells is inside or outside as a whole
			const d2_t a = *reinterpret_cast<const d2_t*>(x + k), b = *reinterpret_cast<const d2_t*>(x + k + 2);
			v[0] = a[0];
			v[1] = a[1];
			v[2] = b[0];
			v[3] = b[1];
		}
		unsigned w[NS];
		nrm_digits4<NS>(v, sh, w);  // non-finite input: caught by K1's sum of squares / K3's flags
		const int64_t ks = k >> 5;
		const int kk = (int)(k & 31);
		char* dst = qrow + ks * 1024 + (((kk >> 4) ^ flip) << 4) + (kk & 15);
#pragma unroll
		for (int s = 0; s < NS; s++) *reinterpret_cast<unsigned*>(dst + s * plane_bytes) = w[s];
#pragma unroll
		for (int s = 0; s < NS - 1; s++) {
			dsum[s] = __builtin_amdgcn_sdot4((int)w[s], 0x01010101, dsum[s], false);
			dsq[s] = (unsigned)__builtin_amdgcn_sdot4((int)w[s], (int)w[s], (int)dsq[s], false);
		}
		// one wave per row: a lane covers cells / 64, so its 32-bit sum of squares (<= 4 * 2^14 per step) is folded into 64 bits
		// every 2^15 steps (2^31 at most), and the 16-lane sums below are taken in 64 bits -- rows of up to 2^22 cells and beyond
		if ((++steps & 0x7fff) == 0) {
#pragma unroll
			for (int s = 0; s < NS - 1; s++) {
				dsq64[s] += dsq[s];
				dsq[s] = 0u;
			}
		}
	}
	if (fix) {  // the row's record for K3's correction and guard (nrm_fix.h)
		double S[5] = {0, 0, 0, 0, 0}, Q2[5] = {0, 0, 0, 0, 0};
#pragma unroll
		for (int s = 0; s < NS - 1; s++) {
			S[s] = (double)wave_sum_i32(dsum[s]);
			unsigned long long q = dsq64[s] + dsq[s];
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
			Q2[s] = (double)q;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) ssq += __shfl_xor(ssq, o, 64);
		if (lane == 0) nrm_fix_record<NS>(fix + row * NRM_FIX_STRIDE, S, Q2, sh, ssq, n_cells);
	}
}

// Operand B as a run of equally sized blocks of a gathered buffer (sharded coex: the partner ranks' digit planes, one block per
// rank, block b at b * stride, visited cyclically from `first`): tile column tj lies in block (first + tj * 128 / rows) % count.
// rows == 0: B is one dense operand.
struct BBlocks {
	int rows;        // padded rows per block (multiple of 128)
	int first, count;
	int64_t stride;  // bytes between blocks
};

// ---- one tile piece: k-steps [k0, k1) of tile (ti, tj) -------------------------------------------------------------
template <int NS>
__device__ __forceinline__ void gram_piece_i8(const char* __restrict__ QA, const char* __restrict__ QB, int64_t plane_a, int64_t plane_b,
											  int64_t nks, const int* __restrict__ ea, const int* __restrict__ eb, double* __restrict__ C,
											  int64_t ldc, int ti, int tj, int k0, int k1, double* __restrict__ slab, int m_rows, int n_rows,
											  int symmetric, int accumulate, BBlocks bb, const char* lds, unsigned lds0) {
	constexpr int STAGE = 8 * NS * 1024;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wid >> 2, wn = wid & 3;  // 2 x 4 waves of 64 rows x 32 columns
	// DMA role: wave `wid` moves the NS digit images of 32-row block (wid & 3) of operand (wid >> 2) of every stage
	const char* qb_t = QB + ((int64_t)tj * 4 * nks) * 1024;  // the tile's 128 rows of B and their exponents
	const int* eb_t = eb + tj * GN;
	if (bb.rows) {
		const int blk = tj * GN / bb.rows, within = tj * GN - blk * bb.rows, wb = (bb.first + blk) % bb.count;
		qb_t = QB + wb * bb.stride + ((int64_t)(within / 32) * nks) * 1024;
		eb_t = eb + wb * bb.rows + within;
	}
	const char* src0 = (wid < 4 ? QA + (((int64_t)ti * 4 + wid) * nks) * 1024 : qb_t + ((int64_t)(wid - 4) * nks) * 1024) + lane * 16;
	const int64_t plane = wid < 4 ? plane_a : plane_b;
	const unsigned dst0 = lds0 + wid * NS * 1024;
	auto issue_one = [&](int buf, int ks, int s) {
		if (QI_EXP & 1) return;
		if (QI_EXP & 2) ks &= 3;
		q_dma16_nc(src0 + s * plane + (int64_t)ks * 1024, dst0 + buf * STAGE + s * 1024);
	};
	auto issue = [&](int buf, int ks) {
#pragma unroll
		for (int s = 0; s < NS; s++) issue_one(buf, ks, s);
	};
	i16_t acc[NS][2];
	auto clear = [&]() {
#pragma unroll
		for (int w = 0; w < NS; w++)
#pragma unroll
			for (int i = 0; i < 2; i++)
#pragma unroll
				for (int j = 0; j < 16; j++) acc[w][i][j] = 0;
	};
	// operand reads: lane l takes the 16 bytes of row l & 31 stored at half (l >> 5) ^ ((row >> 3) & 1)
	const int r = lane & 31;
	const int pos = (2 * r + ((lane >> 5) ^ ((r >> 3) & 1))) * 16;
	const bool diag = symmetric && ti == tj;
	// EDGE tile (round 6, QI_EDGE=1; off by default): a column tile with at most 32 valid columns -- 5000 genes = 39 x 128 + 8: the 39 off-diagonal tiles
	// of the last tile column.  In the usual wave map only waves (0, 0) and (1, 0) have work there, both on SIMD 0, 42 MFMAs each per k-step, and the tile
	// takes as long as a whole one (configs[1] on one box: 4992 genes 2.061 ms, 5000 genes 2.150 ms: 4.3 %).  With this map the four 32-row blocks of the
	// tile go to waves 0-3 -- one per SIMD -- one 32 x 32 sub-tile each (21 MFMAs per k-step, the MASK = 1 copy of the k loop), waves 4-7 only feed the
	// ring: same DMA roles, stages and slab layout.  Parity-green (the Gram engines' tests) and SLOWER: 2.243 ms against 2.150 in three alternating runs on
	// that box (profiles/r06_k2_edge_wave_map.txt) -- a k-step of an edge tile is not bound by its MFMAs (a quarter of a whole tile's per SIMD here) but by
	// what every k-step pays whatever it computes, the stage's 48 KB of DMA through a ring of three and the workgroup barrier, and a compute wave alone on its
	// SIMD has nobody to cover its own DMA issues and operand reads (what round 3's variant (e) found with 42 MFMAs per wave holds with 21).  The bound of ANY
	// edge treatment is the 4992-gene figure; a launch of its own for the edge (transposed, through this kernel) costs 0.136 ms for the 0.09-0.10 it saves
	// (tools/k2_edge_exp.py, profiles/r06_k2_edge_split_launch.txt).
	const bool edge = QI_EDGE && !diag && tj * GN + 32 >= n_rows;
	const int a_blk = edge ? (wid & 3) : wm * 2, b_blk = edge ? 0 : wn;
	const int aoff = a_blk * NS * 1024 + pos;          // + i * NS * 1024 + s * 1024
	const int boff = (4 + b_blk) * NS * 1024 + pos;    // + s * 1024
	// output addressing of this wave's two 32 x 32 tiles: lane holds column lane & 31, rows (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
	const int row_w = ti * GM + a_blk * 32, col_w = tj * GN + b_blk * 32;
	// 32 x 32 sub-tiles that are pure padding, or below the diagonal of a symmetric problem, are neither stored nor -- when both of
	// a wave's are -- computed (5000 genes: the last tile row and column hold 8 valid rows of 128); nobody reads them (K3 sweeps
	// valid rows and, symmetric, the upper triangle), also not through the slabs of split tiles
	bool want[2];
#pragma unroll
	for (int i = 0; i < 2; i++) want[i] = row_w + i * 32 < m_rows && col_w < n_rows && (!diag || col_w + 31 >= row_w + i * 32);
	if (edge) {
		want[0] = want[0] && wid < 4;
		want[1] = false;
	}
	double* cbase;
	int64_t pitch;
	if (slab) {
		cbase = slab + (a_blk * 32) * GN + b_blk * 32;
		pitch = GN;
	} else {
		cbase = C + (int64_t)row_w * ldc + col_w;
		pitch = ldc;
	}
	const int eb_l = eb_t[b_blk * 32 + (lane & 31)];
	auto flush = [&](bool first) {
#pragma unroll
		for (int i = 0; i < 2; i++) {
			if (!want[i]) continue;
#pragma unroll
			for (int q = 0; q < 16; q++) {
				const int rr = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
				// sum_w acc_w 256^w as two exact 64-bit integers (weights 0..2 and 3..NS-1: each below 2^49), so that the one
				// fma that joins them is the only rounding: the chunk's value is the correctly rounded exact integer
				long long lo = 0, hi = 0;
#pragma unroll
				for (int w = 0; w < NS; w++) {
					if (w < 3)
						lo += (long long)acc[w][i][q] << (8 * w);
					else
						hi += (long long)acc[w][i][q] << (8 * (w - 3));
				}
				double v = fma((double)hi, 16777216.0, (double)lo);
				v = ldexp(v, ea[row_w + rr] + eb_l + 8 * (NS - 1));
				double* o = cbase + (int64_t)rr * pitch + (lane & 31);
				*o = (first && !(accumulate && !slab)) ? v : *o + v;
			}
		}
	};
	clear();
	__syncthreads();  // previous piece done with LDS
	if (k0 >= k1) return;  // (the schedule has no empty pieces)
	issue(0, k0);
	issue(1, min(k0 + 1, k1 - 1));
	int in_chunk = 0;
	bool first = true;
	// two copies of the k loop (compute / only feed the ring), chosen per piece: a branch INSIDE the loop makes the compiler shuffle the 192 accumulator registers
	// on every iteration (26 ms instead of 2.2 on C2)
	auto kloop = [&](auto mask) {
	constexpr int MASK = decltype(mask)::value;  // which of the wave's two 32-row halves are computed
	for (int ks = k0; ks < k1; ks++) {
		const int buf = (ks - k0) % QD;
		asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NS) : "memory");  // this wave's images of stage ks have landed (stage ks + 1 may be in flight)
		if (!(QI_EXP & 4)) __syncthreads();  // everyone's have; all waves are done reading stage ks - 1, whose buffer is refilled now
		// The k-step by hand: row s of the digit-pair triangle (pairs (s, t >= NS-1-s): 2 (s + 1) MFMAs) needs fa[s] and fb[NS-1-s]
		// for the first time; they are read one row ahead, and ONE refill DMA of stage ks + 2 is issued after each row (the last in
		// the middle of the longest row).  A DMA instruction costs its wave 60-190 issue cycles; bunched behind the barrier with
		// the 18 operand reads -- as this loop was at first -- neither wave of the SIMD feeds the matrix core meanwhile; spread out,
		// each falls into the 32 cycles the partner wave's MFMA holds the pipe anyway.  sched_barrier pins the order.
		const char* st = lds + buf * STAGE;
		// past the end of the piece the refill re-fetches the last stage into the free buffer: no branch in the MFMA stream, and the
		// count of DMAs in flight is the same at every step (drained after the loop)
		const int nbuf = (ks - k0 + 2) % QD, nks = min(ks + 2, k1 - 1);
		i4_t fa[NS][2], fb[NS];
		auto read_row = [&](int s) {
#pragma unroll
			for (int i = 0; i < 2; i++)
				if (MASK >> i & 1) fa[s][i] = *reinterpret_cast<const i4_t*>(st + aoff + (i * NS + s) * 1024);
			fb[NS - 1 - s] = *reinterpret_cast<const i4_t*>(st + boff + (NS - 1 - s) * 1024);
		};
		auto mfma_row = [&](int s, int t0, int t1) {
#pragma unroll
			for (int t = t0; t < t1; t++)
#pragma unroll
				for (int i = 0; i < 2; i++)
					if (MASK >> i & 1)
						acc[s + t - (NS - 1)][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s][i], fb[t], acc[s + t - (NS - 1)][i], 0, 0, 0);
		};
		auto dma = [&](int j) {
			__builtin_amdgcn_sched_barrier(0);
			issue_one(nbuf, nks, j);
			__builtin_amdgcn_sched_barrier(0);
		};
		if (MASK) {
			read_row(0);
			read_row(1);
#pragma unroll
			for (int s = 0; s < NS - 1; s++) {
				mfma_row(s, NS - 1 - s, NS);
				dma(s);
				if (s + 2 < NS) read_row(s + 2);
			}
			mfma_row(NS - 1, 0, NS / 2);
			dma(NS - 1);
			mfma_row(NS - 1, NS / 2, NS);
		} else {  // this wave only feeds the ring
#pragma unroll
			for (int s = 0; s < NS; s++) dma(s);
		}
		if (MASK && ++in_chunk == QCHUNK && ks + 1 < k1) {  // int32 headroom used up: combine in fp64, start a new chunk
			flush(first);
			first = false;
			clear();
			in_chunk = 0;
		}
	}
	};
	if (QI_EDGE && want[0] && !want[1])  // the first half only: the waves of an edge tile (above), and the last tile row when it holds at most 32 valid rows
		kloop(std::integral_constant<int, 1>{});
	else if (want[0] || want[1])  // (the second half alone -- a diagonal tile's lower-left wave -- is computed with both: no fourth copy of the loop)
		kloop(std::integral_constant<int, 3>{});
	else
		kloop(std::integral_constant<int, 0>{});
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the refills issued past the end: nothing may land after the workgroup has gone
	flush(first);
}

template <int NS>
__global__ void __launch_bounds__(512) k_gram_i8(const char* __restrict__ QA, const char* __restrict__ QB, int64_t plane_a, int64_t plane_b,
												 int64_t nks, const int* __restrict__ ea, const int* __restrict__ eb, double* __restrict__ C,
												 int64_t ldc, int symmetric, GramSched s, BBlocks bb) {
	__shared__ __attribute__((aligned(1024))) char lds[QD * 8 * NS * 1024];
	typedef __attribute__((address_space(3))) char* lds_ptr_t;
	const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)lds);
	gram_for_each_piece(s, [&](int t, int k0, int k1, double* slab) {
		int ti, tj;
		gram_tile_coords(s.tile0 + t, symmetric, s.ntm, s.ntn, ti, tj);
		gram_piece_i8<NS>(QA, QB, plane_a, plane_b, nks, ea, eb, C, ldc, ti, tj, k0, k1, slab, s.m_rows, s.n_rows, symmetric, s.accumulate, bb, lds, lds0);
	});
}

// ---- C ABI -----------------------------------------------------------------------------------------------------------
static int g_num_cu_q = 0;

extern "C" int64_t nrm_quant_bytes(int64_t rows_pad, int64_t k_pad, int nslices) {
	if (rows_pad < 0 || k_pad < 0 || nslices < 1) return 0;
	const int64_t nks = (k_pad + QK - 1) / QK;
	return (rows_pad / 32) * nks * 1024 * nslices;
}

extern "C" int nrm_quantize_rows(const double* d_x, int64_t rows_pad, int64_t k_pad, int64_t ldx, int nslices, void* d_q, int32_t* d_exp,
								 double* d_fix, int64_t n_cells, void* stream) {
	NRM_REQUIRE(nslices == 5 || nslices == 6, "nrm_quantize_rows: 5 or 6 slices");
	NRM_REQUIRE(rows_pad >= 0 && rows_pad % 32 == 0 && k_pad > 0 && k_pad % 16 == 0 && ldx >= k_pad && ldx % 2 == 0,
				"nrm_quantize_rows: rows must be padded to 32 (%d for nrm_gram_i8), cells to 16", GM);
	if (rows_pad == 0) return NRM_OK;
	NRM_REQUIRE(d_x && d_q && d_exp && (uintptr_t)d_x % 16 == 0 && (uintptr_t)d_q % 16 == 0, "nrm_quantize_rows: null or misaligned pointer");
	NRM_REQUIRE(!d_fix || (n_cells > 0 && n_cells <= k_pad && k_pad < (1 << 22)), "nrm_quantize_rows: row records need 0 < n_cells <= k_pad < 2^22");
	const int64_t nks = (k_pad + QK - 1) / QK;
	const int64_t plane = (rows_pad / 32) * nks * 1024;
	dim3 grid((unsigned)(rows_pad / 4));
	if (nslices == 5)
		hipLaunchKernelGGL(k_quantize_rows<5>, grid, dim3(256), 0, (hipStream_t)stream, d_x, k_pad, ldx, (char*)d_q, plane, nks, d_exp, d_fix, (double)n_cells);
	else
		hipLaunchKernelGGL(k_quantize_rows<6>, grid, dim3(256), 0, (hipStream_t)stream, d_x, k_pad, ldx, (char*)d_q, plane, nks, d_exp, d_fix, (double)n_cells);
	return nrm_check_launch("k_quantize_rows");
}

static int gram_i8_impl(const void* d_qa, const int32_t* d_ea, int64_t plane_a_bytes, const void* d_qb, const int32_t* d_eb,
						int64_t plane_b_bytes, int64_t m_pad, int64_t n_pad, int64_t k_pad, int nslices, double* d_dot, int64_t ldd,
						int symmetric, int64_t m_rows, int64_t n_rows, int64_t row0, int64_t row1, int accumulate, BBlocks bb, void* d_work, void* stream) {
	NRM_REQUIRE(nslices == 5 || nslices == 6, "nrm_gram_i8: 5 or 6 slices");
	NRM_REQUIRE(m_pad >= 0 && n_pad >= 0 && k_pad > 0 && m_pad % GM == 0 && n_pad % GN == 0, "nrm_gram_i8: sizes must be padded to %d", GM);
	NRM_REQUIRE(ldd >= n_pad && ldd % 2 == 0, "nrm_gram_i8: pitch too small");
	NRM_REQUIRE(!symmetric || m_pad == n_pad, "nrm_gram_i8: symmetric needs square output");
	NRM_REQUIRE(row0 >= 0 && row0 <= row1 && row1 <= m_pad && row0 % (GSB * GM) == 0 && (row1 % (GSB * GM) == 0 || row1 == m_pad),
				"nrm_gram_i8_band: rows [row0, row1) must be cut at multiples of %d", GSB * GM);
	if (m_pad == 0 || n_pad == 0 || row0 == row1) return NRM_OK;
	NRM_REQUIRE(d_qa && d_qb && d_ea && d_eb && d_dot && d_work, "nrm_gram_i8: null pointer");
	if (g_num_cu_q == 0) {
		int dev = 0;
		NRM_HIP(hipGetDevice(&dev));
		NRM_HIP(hipDeviceGetAttribute(&g_num_cu_q, hipDeviceAttributeMultiprocessorCount, dev));
		if (g_num_cu_q <= 0) g_num_cu_q = 256;
	}
	const int64_t nks = (k_pad + QK - 1) / QK;
	// distance between digit planes: dense by default; larger when the operand is a block of rows of a bigger quantised matrix
	const int64_t plane_a = plane_a_bytes ? plane_a_bytes : (m_pad / 32) * nks * 1024,
				  plane_b = plane_b_bytes ? plane_b_bytes : ((bb.rows ? bb.rows : n_pad) / 32) * nks * 1024;
	NRM_REQUIRE(plane_a >= (m_pad / 32) * nks * 1024 && plane_b >= ((bb.rows ? bb.rows : n_pad) / 32) * nks * 1024,
				"nrm_gram_i8: plane pitch smaller than the operand");
	GramSched s;
	NRM_TRY_RC(gram_plan(s, m_pad, n_pad, nks, symmetric, m_rows, n_rows, row0, row1, g_num_cu_q, (double*)d_work));  // one workgroup per CU
	s.accumulate = accumulate ? 1 : 0;
	if (nslices == 5)
		hipLaunchKernelGGL(k_gram_i8<5>, dim3((unsigned)s.nwg), dim3(512), 0, (hipStream_t)stream, (const char*)d_qa, (const char*)d_qb, plane_a,
						   plane_b, nks, d_ea, d_eb, d_dot, ldd, symmetric, s, bb);
	else
		hipLaunchKernelGGL(k_gram_i8<6>, dim3((unsigned)s.nwg), dim3(512), 0, (hipStream_t)stream, (const char*)d_qa, (const char*)d_qb, plane_a,
						   plane_b, nks, d_ea, d_eb, d_dot, ldd, symmetric, s, bb);
	if (s.tiles_al + s.tiles_sk > 0)
		hipLaunchKernelGGL(k_gram_fixup<1>, dim3((unsigned)(s.tiles_al + s.tiles_sk), GM / GFIX_ROWS), dim3(256), 0, (hipStream_t)stream, d_dot, ldd, symmetric, s);
	return nrm_check_launch("k_gram_i8");
}

extern "C" int nrm_gram_i8_band(const void* d_qa, const int32_t* d_ea, int64_t plane_a_bytes, const void* d_qb, const int32_t* d_eb,
								int64_t plane_b_bytes, int64_t m_pad, int64_t n_pad, int64_t k_pad, int nslices, double* d_dot, int64_t ldd,
								int symmetric, int64_t m_rows, int64_t n_rows, int64_t row0, int64_t row1, void* d_work, void* stream) {
	return gram_i8_impl(d_qa, d_ea, plane_a_bytes, d_qb, d_eb, plane_b_bytes, m_pad, n_pad, k_pad, nslices, d_dot, ldd, symmetric, m_rows, n_rows, row0,
						row1, 0, BBlocks{0, 0, 1, 0}, d_work, stream);
}

// One cell chunk of a contraction whose operands arrive in pieces along the cells (sharded coex: the digit planes of the other
// ranks' blocks travel chunk by chunk): accumulate != 0 adds this chunk's exact partial dot products to d_dot in fp64.
// b_block_rows != 0: operand B is b_blocks consecutive blocks (cyclically from b_first, of b_count) of a gathered buffer, each a
// dense quantised operand of b_block_rows padded rows at d_qb + block * b_block_stride_bytes with its exponents at
// d_eb + block * b_block_rows; n_pad = b_blocks * b_block_rows -- all full partner blocks of a rank in ONE launch.
extern "C" int nrm_gram_i8_chunk(const void* d_qa, const int32_t* d_ea, int64_t plane_a_bytes, const void* d_qb, const int32_t* d_eb,
								 int64_t plane_b_bytes, int64_t m_pad, int64_t n_pad, int64_t k_pad, int nslices, double* d_dot, int64_t ldd,
								 int symmetric, int64_t m_rows, int64_t n_rows, int accumulate, int64_t b_block_rows, int64_t b_block_stride_bytes,
								 int b_first, int b_count, void* d_work, void* stream) {
	BBlocks bb{0, 0, 1, 0};
	if (b_block_rows) {
		NRM_REQUIRE(!symmetric && b_block_rows > 0 && b_block_rows % GN == 0 && n_pad % b_block_rows == 0 && b_count > 0 && b_first >= 0 &&
						b_first < b_count && n_pad / b_block_rows <= b_count && b_block_stride_bytes % 16 == 0 && b_block_rows < (1 << 30),
					"nrm_gram_i8_chunk: bad block description of operand B");
		bb = BBlocks{(int)b_block_rows, b_first, b_count, b_block_stride_bytes};
	}
	return gram_i8_impl(d_qa, d_ea, plane_a_bytes, d_qb, d_eb, plane_b_bytes, m_pad, n_pad, k_pad, nslices, d_dot, ldd, symmetric, m_rows, n_rows, 0,
						m_pad, accumulate, bb, d_work, stream);
}
