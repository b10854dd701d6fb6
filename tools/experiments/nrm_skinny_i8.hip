// K2s on the int8 matrix cores: the streaming de kernel (nrm_gram_skinny.hip: association.py:224-235 for n_x + n_cov <= 32, every
// expression value read once) with the contraction in exact fixed point, like K2 (nrm_gram_i8.hip), instead of on the fp64 matrix
// cores, whose 20 Z rows are the floor of the fp64 formulation (1.56 ms alone on BASELINE configs[2] against 1.3 ms of HBM time).
//
//     G[y, :] = sum_k Y[y,k] Z[:,k]   (ny, 32)      ss[y] = sum_k Y[y,k]^2 (fp64, vector ALU)      dig[y, s] = sum_k d_s(Y[y,k])
// (the digit sums come from the matrix cores too: the caller sets Z row 31 to a constant, whose fixed-point image is 32 * 256^5 --
// only the top digit -- so that column 31's accumulator of weight s holds 32 * sum_k d_s and nothing else)
//
// The expression rows are RAW (fp32 / fp64) in HBM and are turned into 46-bit fixed point ON THE FLY: tools/quant_probe.hip measured
// that the conversion rides under the stream (1.31 ms against 1.28 ms for the bare 8 GB).  What that needs is every row's scale
// BEFORE the row is streamed: ysh[y] with |Y[y,k]| < 2^(ysh + 46), from nrm_row_scales -- one extra pass, worth it when the same
// resident rows are streamed more than once (a DePlan's second step on; the caller keeps the scales with the plan).
//
// STATUS (round 3): correct (tests/test_gpu_round3.py, tools/k2s_i8_check.py: G to 4e-15 of |y||z|, sums of squares and digit sums
// exact) but NOT the default -- on configs[2] the form below (rows through registers) takes 4.7 ms, the DMA form further down 2.5 ms,
// the fp64 kernel 1.87 ms.  The arithmetic is not the problem (conversion + matrix cores fed zeros: 0.95 ms, against a matrix-core
// floor of 1.56 ms in fp64); feeding it is: see the ablations in DESIGN.md section 4.  Opt in: NRM_DE_I8=1 (NRM_DE_I8_DMA=0: this form
// for fp32 rows too).
//
// Structure: one workgroup per CU owns 256 rows; 4 LOADER waves pull the rows through registers (two stages of 32 cells in
// flight), cut every value into 6 balanced base-256 digits (nrm_digits.h) and write them to LDS in K2's operand layout (1 KB
// images of 32 rows x 32 bytes, halves swapped when (row >> 3) & 1), the 6 digit planes of Z next to them; 4 MFMA waves of 64
// rows run K2's k-step on them -- 21 digit pairs, 42 v_mfma_i32_32x32x32_i8, one int32 accumulator set per weight -- and flush to
// fp64 every 512 k-steps.  One barrier per k-step; LDS 2 x (48 + 6) KB.  Sums of squares stay on the vector ALU in fp64 (they
// must be exact: |y~|^2 = |y|^2 - ...); the digit sums feed K3's exact correction for the dropped digit products (nrm_fix.h).
// Persistent DP + stream-K schedule and deterministic fix-up as in nrm_gram_skinny.hip.
#include <cstdlib>
#include "nrm_common.h"
#include "nrm_digits.h"

#define SQN 32        // columns of G (= rows of Z)
#define SQC 128       // cells per schedule unit
#define SQ_TM 256     // rows per workgroup tile
#define SQ_NS 6       // digit planes of both operands
#define SQ_DIG 8      // doubles per row in the digit-sum output (5 used: planes 0..4)
#define SQ_FLUSH 512  // k-steps per int32 accumulation chunk (16 384 cells)
#ifndef SQ_EXP
#define SQ_EXP 0  // timing experiments (tools/build_exp.sh): 1 no MFMAs, 2 no digit conversion (raw words to LDS), 4 no HBM loads, 8 no LDS writes
#endif

typedef int i4_t __attribute__((ext_vector_type(4)));
typedef int i16_t __attribute__((ext_vector_type(16)));
typedef double d2v_t __attribute__((ext_vector_type(2)));

struct SkinnyQSched {
	int nkt, tiles_dp, tiles_sk, units_per_wg, nwg;
	int tm;        // rows per workgroup tile (256: rows through registers; 128: rows by DMA)
	double* work;  // per partial piece: (256 x 32) G, 256 sums of squares, (256 x 8) digit sums; two pieces per workgroup
};
#define SQ_SLAB ((int64_t)SQ_TM * (SQN + 1 + SQ_DIG))

template <typename T>
struct Row4;  // four consecutive cells of an expression row, as loaded
template <>
struct Row4<float> {
	float4 v;
	__device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
	__device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
	__device__ __forceinline__ void get(double (&x)[4]) const {
		x[0] = v.x;
		x[1] = v.y;
		x[2] = v.z;
		x[3] = v.w;
	}
};
template <>
struct Row4<double> {
	d2v_t a, b;
	__device__ __forceinline__ void load(const double* p) {
		a = *reinterpret_cast<const d2v_t*>(p);
		b = *reinterpret_cast<const d2v_t*>(p + 2);
	}
	__device__ __forceinline__ void zero() { a = b = (d2v_t){0.0, 0.0}; }
	__device__ __forceinline__ void get(double (&x)[4]) const {
		x[0] = a[0];
		x[1] = a[1];
		x[2] = b[0];
		x[3] = b[1];
	}
};

template <typename T>
__global__ void __launch_bounds__(512) k_skinny_i8(const T* __restrict__ Y, int64_t rows, int64_t n, int64_t ldy, const int* __restrict__ ysh,
												   const char* __restrict__ ZQ, int64_t zplane, const int* __restrict__ zsh, double* __restrict__ G,
												   double* __restrict__ ss, double* __restrict__ dig, SkinnyQSched s) {
	constexpr int NS = SQ_NS;
	constexpr int ASTAGE = (SQ_TM / 32) * NS * 1024, ZSTAGE = NS * 1024;
	__shared__ __attribute__((aligned(1024))) char alds[2 * ASTAGE];
	__shared__ __attribute__((aligned(1024))) char zlds[2 * ZSTAGE];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
	const bool loader = wid >= 4;
	const int per_xcd = s.nwg >> 3;
	const int p = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
	int t_dp = p;
	int64_t u = (int64_t)p * s.units_per_wg;
	const int64_t total = (int64_t)s.tiles_sk * s.nkt;
	int64_t uend = u + s.units_per_wg;
	if (uend > total) uend = total;
	int sk_piece = 0;
	const int nks_all = (int)((n + 31) / 32);  // k-steps that hold cells of the rows (Z is padded beyond)
	const int64_t n16 = (n + 15) / 16 * 16;
	// loader geometry: wave lw, lane L -> rows lw*64 + 8 j + (L >> 3), j = 0..7; cells 4 (L & 7) .. + 3 of the k-step
	const int lw = wid - 4, lr = lane >> 3, c4 = (lane & 7) * 4;
	// MFMA geometry (K2's): lane l reads the 16 bytes of row l & 31 stored at half (l >> 5) ^ ((row >> 3) & 1)
	const int mr = lane & 31;
	const int pos = (2 * mr + ((lane >> 5) ^ ((mr >> 3) & 1))) * 16;
	for (;;) {
		int t, c0, c1;
		double* slab = nullptr;
		if (t_dp < s.tiles_dp) {
			t = t_dp;
			c0 = 0;
			c1 = s.nkt;
			t_dp += s.nwg;
		} else if (u < uend) {
			const int ts = (int)(u / s.nkt);
			c0 = (int)(u - (int64_t)ts * s.nkt);
			int64_t c1l = c0 + (uend - u);
			c1 = c1l > s.nkt ? s.nkt : (int)c1l;
			t = s.tiles_dp + ts;
			u += c1 - c0;
			if (!(c0 == 0 && c1 == s.nkt)) slab = s.work + ((int64_t)2 * p + sk_piece) * SQ_SLAB;
			sk_piece++;
		} else {
			break;
		}
		const int st0 = c0 * (SQC / 32);
		int st1 = c1 * (SQC / 32);
		if (st1 > nks_all) st1 = nks_all;
		const int64_t row0 = (int64_t)t * SQ_TM;
		__syncthreads();  // previous piece: everyone is done with LDS
		if (loader) {
			// ---- loader waves: HBM -> registers -> digits -> LDS; sums of squares and digit sums of their rows ----
			// rows lw*64 + 8 j + lr, j = 0..7: one base pointer, 8 ldy elements apart; rows past the end contribute zeros
			const int64_t rbase = row0 + lw * 64 + lr;
			const T* src0 = Y + (rbase < rows ? rbase : 0) * ldy + c4;
			const int64_t jstride = 8 * ldy;
			int sh[8];
			double sq[8];
			unsigned live = 0;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const int64_t r = rbase + 8 * j;
				if (r < rows) live |= 1u << j;
				sh[j] = ysh[r < rows ? r : 0];
				sq[j] = 0.0;
			}
			// this lane's share of a Z stage (6 KB = 384 chunks of 16 bytes): chunk lw*64 + lane of planes 0-3, and for lw < 2 plane 4 + lw
			const char* zsrc0 = ZQ + (int64_t)lw * zplane + lane * 16;
			const char* zsrc1 = ZQ + (int64_t)(4 + (lw & 1)) * zplane + lane * 16;
			struct Stage {
				Row4<T> y[8];
				i4_t z0, z1;
			};
			auto issue = [&](Stage& g, int ks) {
				const bool inside = (int64_t)ks * 32 + c4 < n16;  // rows are readable, and zero, from n up to n16 = round_up(n, 16)
#pragma unroll
				for (int j = 0; j < 8; j++) {
					if (inside && (live >> j & 1) && !(SQ_EXP & 4))
						g.y[j].load(src0 + j * jstride + (int64_t)ks * 32);
					else
						g.y[j].zero();
				}
				g.z0 = *reinterpret_cast<const i4_t*>(zsrc0 + (int64_t)ks * 1024);
				if (lw < 2) g.z1 = *reinterpret_cast<const i4_t*>(zsrc1 + (int64_t)ks * 1024);
			};
			auto process = [&](const Stage& g, int buf) {
				char* a = alds + buf * ASTAGE;
#pragma unroll
				for (int j = 0; j < 8; j++) {
					double x[4];
					g.y[j].get(x);
#pragma unroll
					for (int i = 0; i < 4; i++) sq[j] = fma(x[i], x[i], sq[j]);
					unsigned w[NS];
					if (SQ_EXP & 2) {
#pragma unroll
						for (int q = 0; q < NS; q++) w[q] = (unsigned)__double2loint(x[q & 3]) + q;
					} else
						nrm_digits4<NS>(x, sh[j], w);
					// row lw*64 + 8 j + lr of the tile: block lw*2 + (j >> 2), row (j & 3)*8 + lr of it, halves swapped when j & 1
					const int rr = (j & 3) * 8 + lr;
					char* dst = a + ((lw * 2 + (j >> 2)) * NS) * 1024 + (2 * rr) * 16 + (((c4 >> 4) ^ (j & 1)) << 4) + (c4 & 15);
					if (SQ_EXP & 8) {
						if (w[0] == 0x12345u && w[5] == 0x54321u) *reinterpret_cast<unsigned*>(dst) = w[1] ^ w[2] ^ w[3] ^ w[4];
					} else {
#pragma unroll
						for (int q = 0; q < NS; q++) *reinterpret_cast<unsigned*>(dst + q * 1024) = w[q];
					}
				}
				char* z = zlds + buf * ZSTAGE;
				*reinterpret_cast<i4_t*>(z + lw * 1024 + lane * 16) = g.z0;
				if (lw < 2) *reinterpret_cast<i4_t*>(z + (4 + lw) * 1024 + lane * 16) = g.z1;
			};
			// Two register sets: while one stage is converted, the next is on its way from HBM.  (Measured on configs[2], 8 GB: this loop
			// alone -- loads and LDS writes, nothing else -- takes 2.1 ms = 3.8 TB/s: one stage per loader wave is 32 KB in flight per CU,
			// and a third register set spills (2.7 ms).  The conversion and the matrix cores alone, fed zeros, take 0.95 ms.  See DESIGN.md.)
			Stage g0, g1;
			if (st0 < st1) issue(g0, st0);
			if (st0 + 1 < st1) issue(g1, st0 + 1);
			if (st0 < st1) process(g0, 0);
			__syncthreads();  // stage st0 is in LDS
			for (int ks = st0; ks < st1; ks += 2) {
				// k-step ks (the MFMA waves contract buffer 0): fetch stage ks + 2, convert stage ks + 1 into buffer 1
				if (ks + 2 < st1) issue(g0, ks + 2);
				if (ks + 1 < st1) process(g1, 1);
				__syncthreads();
				if (ks + 1 < st1) {  // k-step ks + 1 (buffer 1): fetch stage ks + 3, convert stage ks + 2 into buffer 0
					if (ks + 3 < st1) issue(g1, ks + 3);
					if (ks + 2 < st1) process(g0, 0);
					__syncthreads();
				}
			}
			// the rows' sums: 8 lanes share a row
			double* sbase = slab ? slab + SQ_TM * SQN : ss + row0;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				double v = sq[j];
				v += __shfl_xor(v, 1, 64);
				v += __shfl_xor(v, 2, 64);
				v += __shfl_xor(v, 4, 64);
				if ((lane & 7) == 0) sbase[lw * 64 + 8 * j + lr] = v;
			}
			continue;
		}
		// ---- MFMA waves: K2's k-step on the digit images in LDS ----
		i16_t acc[NS][2];
		auto clear = [&]() {
#pragma unroll
			for (int w = 0; w < NS; w++)
#pragma unroll
				for (int i = 0; i < 2; i++)
#pragma unroll
					for (int j = 0; j < 16; j++) acc[w][i][j] = 0;
		};
		double* gbase = slab ? slab + (wid * 64) * SQN : G + (row0 + wid * 64) * SQN;
		double* dbase = slab ? slab + SQ_TM * (SQN + 1) + (wid * 64) * SQ_DIG : dig + (row0 + wid * 64) * SQ_DIG;
		const int zsh_l = zsh[lane & 31];
		auto flush = [&](bool first) {
#pragma unroll
			for (int i = 0; i < 2; i++)
#pragma unroll
				for (int q = 0; q < 16; q++) {
					const int rr = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
					long long lo = 0, hi = 0;
#pragma unroll
					for (int w = 0; w < NS; w++) {
						if (w < 3)
							lo += (long long)acc[w][i][q] << (8 * w);
						else
							hi += (long long)acc[w][i][q] << (8 * (w - 3));
					}
					double v = fma((double)hi, 16777216.0, (double)lo);  // the exact integer, rounded once
					const int64_t r = row0 + wid * 64 + rr;
					v = ldexp(v, ysh[r < rows ? r : 0] + zsh_l + 8 * (NS - 1));
					double* o = gbase + (int64_t)rr * SQN + (lane & 31);
					*o = first ? v : *o + v;
					if ((lane & 31) == SQN - 1) {  // the digit-sum column: weight s of it is 32 sum_k d_s alone (see the header)
#pragma unroll
						for (int w = 0; w < SQ_DIG; w++) {
							const double dv = w < NS - 1 ? (double)acc[w][i][q] * 0.03125 : 0.0;
							double* od = dbase + (int64_t)rr * SQ_DIG + w;
							*od = first ? dv : *od + dv;
						}
					}
				}
		};
		clear();
		int in_chunk = 0;
		bool first = true;
		const int aoff = (wid * 2) * NS * 1024 + pos;
		auto kstep = [&](int buf) {
			const char* a = alds + buf * ASTAGE + aoff;
			const char* z = zlds + buf * ZSTAGE + pos;
			i4_t fa[NS][2], fb[NS];
			auto read_row = [&](int sI) {
#pragma unroll
				for (int i = 0; i < 2; i++) fa[sI][i] = *reinterpret_cast<const i4_t*>(a + (i * NS + sI) * 1024);
				fb[NS - 1 - sI] = *reinterpret_cast<const i4_t*>(z + (NS - 1 - sI) * 1024);
			};
			if (SQ_EXP & 1) return;
			read_row(0);
			read_row(1);
#pragma unroll
			for (int sI = 0; sI < NS; sI++) {
#pragma unroll
				for (int tI = NS - 1 - sI; tI < NS; tI++)
#pragma unroll
					for (int i = 0; i < 2; i++)
						acc[sI + tI - (NS - 1)][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[sI][i], fb[tI], acc[sI + tI - (NS - 1)][i], 0, 0, 0);
				if (sI + 2 < NS) read_row(sI + 2);
			}
		};
		__syncthreads();  // stage st0 is in LDS
		for (int ks = st0; ks < st1; ks += 2) {
			kstep(0);
			__syncthreads();
			if (ks + 1 < st1) {
				kstep(1);
				__syncthreads();
			}
			in_chunk += 2;
			if (in_chunk >= SQ_FLUSH && ks + 2 < st1) {  // int32 headroom used up: combine in fp64, start a new chunk
				flush(first);
				first = false;
				clear();
				in_chunk = 0;
			}
		}
		if (st0 < st1)
			flush(first);
		else if (slab) {  // (a piece beyond the last cells of the rows: its slab is summed by the fix-up all the same)
			clear();
			flush(true);
		}
	}
}

// ---- the same with the raw rows brought in by DMA (fp32 rows) ----------------------------------------------------------------------
// The kernel above pulls the rows through the registers of four loader waves, which cannot keep the HBM pipe full.  Here, as in
// the fp64 kernel, LOADER waves do nothing but issue global -> LDS DMA into a ring of raw stages (no registers, three stages ahead),
// CONVERTER waves read a raw stage from LDS, cut it into digits and write the digit images, MFMA waves contract them.  128-row
// tiles so that the raw ring (4 x 16 KB), two digit stages (2 x 24 KB) and Z (2 x 6 KB) fit in LDS; 12 waves, 3 per SIMD: one
// loader, one converter (32 rows), one MFMA wave (32 rows x 32 Z rows, 96 accumulators) each.  One barrier per k-step:
//   after it, raw stage ks + 1 has landed and digit buffer ks % 2 holds stage ks; during k-step ks the MFMA waves contract that
//   buffer, the converters turn raw stage ks + 1 into the other one, the loaders refill the raw buffer of stage ks (converted during
//   k-step ks - 1) with stage ks + 4 and wait for stage ks + 2.
#define SD_TM 128
#define SD_D 4
__device__ __forceinline__ void sd_dma16(const void* gsrc, unsigned lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
				 : "=&s"(keep)
				 : "v"(gsrc), "s"(lds_dst)
				 : "memory");
}
#define SD_PS 6  // DMA instructions per loader wave and stage: 4 of the 16 row groups (8 rows x 128 B each) + 2 of the 8 Z slots (6 planes, 2 spares)
__device__ __forceinline__ void sd_wait(int younger) {  // all of this wave's DMA done except those of `younger` (0..3) stages
	if (younger <= 0)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	else if (younger == 1)
		asm volatile("s_waitcnt vmcnt(%0)" ::"i"(SD_PS) : "memory");
	else if (younger == 2)
		asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * SD_PS) : "memory");
	else
		asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * SD_PS) : "memory");
}

#define SD_SLAB ((int64_t)SD_TM * (SQN + 1 + SQ_DIG))

__global__ void __launch_bounds__(768) k_skinny_i8_dma(const float* __restrict__ Y, int64_t rows, int64_t n, int64_t ldy, const int* __restrict__ ysh,
													   const char* __restrict__ ZQ, int64_t zplane, const int* __restrict__ zsh, double* __restrict__ G,
													   double* __restrict__ ss, double* __restrict__ dig, SkinnyQSched s) {
	constexpr int NS = SQ_NS, D = SD_D;
	constexpr int RSTAGE = SD_TM * 128, ASTAGE = (SD_TM / 32) * NS * 1024, ZSTAGE = NS * 1024, ZD = D + 1;
	__shared__ __attribute__((aligned(1024))) char rlds[D * RSTAGE];
	__shared__ __attribute__((aligned(1024))) char alds[2 * ASTAGE];
	// the Z planes of a stage travel with its rows, D stages ahead, and are read at its k-step: D + 1 buffers (+ 2 KB for the two spare slots)
	__shared__ __attribute__((aligned(1024))) char zlds[ZD * ZSTAGE + 2048];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0-3 MFMA, 4-7 converters, 8-11 loaders
	typedef __attribute__((address_space(3))) char* lds_ptr_t;
	const unsigned rlds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)rlds);
	const unsigned zlds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)zlds);
	const int per_xcd = s.nwg >> 3;
	const int p = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
	int t_dp = p;
	int64_t u = (int64_t)p * s.units_per_wg;
	const int64_t total = (int64_t)s.tiles_sk * s.nkt;
	int64_t uend = u + s.units_per_wg;
	if (uend > total) uend = total;
	int sk_piece = 0;
	const int nks_all = (int)((n + 31) / 32);
	const int64_t n16 = (n + 15) / 16 * 16;
	const int mr = lane & 31;
	const int pos = (2 * mr + ((lane >> 5) ^ ((mr >> 3) & 1))) * 16;
	for (;;) {
		int t, c0, c1;
		double* slab = nullptr;
		if (t_dp < s.tiles_dp) {
			t = t_dp;
			c0 = 0;
			c1 = s.nkt;
			t_dp += s.nwg;
		} else if (u < uend) {
			const int ts = (int)(u / s.nkt);
			c0 = (int)(u - (int64_t)ts * s.nkt);
			int64_t c1l = c0 + (uend - u);
			c1 = c1l > s.nkt ? s.nkt : (int)c1l;
			t = s.tiles_dp + ts;
			u += c1 - c0;
			if (!(c0 == 0 && c1 == s.nkt)) slab = s.work + ((int64_t)2 * p + sk_piece) * SD_SLAB;
			sk_piece++;
		} else {
			break;
		}
		const int st0 = c0 * (SQC / 32);
		int st1 = c1 * (SQC / 32);
		if (st1 > nks_all) st1 = nks_all;
		const int64_t row0 = (int64_t)t * SD_TM;
		__syncthreads();  // previous piece: everyone is done with LDS
		if (wid >= 8) {
			// ---- loader waves: DMA issue, counted waits, the barriers ----
			const int li = wid - 8;
			const float* ysrc[4];  // row groups li, li + 4, li + 8, li + 12 (8 rows each): lane -> row lane >> 3, 16-byte chunk lane & 7
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int64_t r = row0 + 8 * (li + 4 * j) + (lane >> 3);
				ysrc[j] = Y + (r < rows ? r : 0) * ldy + (lane & 7) * 4;
			}
			const char* zsrc[2];  // Z slots li and li + 4 (slots 6 and 7 re-fetch plane 0 into spare LDS: every loader issues the same count)
#pragma unroll
			for (int q = 0; q < 2; q++) {
				const int slot = li + 4 * q;
				zsrc[q] = ZQ + (int64_t)(slot < NS ? slot : 0) * zplane + lane * 16;
			}
			auto issue = [&](int buf, int ks) {
				const int64_t k0 = (int64_t)ks * 32;
				const bool past = k0 + (lane & 7) * 4 >= n16;  // beyond the readable part of the rows: fetch anything valid, the converter zeroes it
				if (SQ_EXP & 4) return;
#pragma unroll
				for (int j = 0; j < 4; j++) sd_dma16(past ? (const void*)Y : (const void*)(ysrc[j] + k0), rlds0 + buf * RSTAGE + (li + 4 * j) * 1024);
				const unsigned zb = zlds0 + ((ks - st0) % ZD) * ZSTAGE;
				sd_dma16(zsrc[0] + (int64_t)ks * 1024, zb + li * 1024);  // planes 0-3
				sd_dma16(zsrc[1] + (int64_t)ks * 1024, li < 2 ? zb + (li + 4) * 1024 : zlds0 + ZD * ZSTAGE + (li - 2) * 1024);  // planes 4, 5; spares
			};
			for (int d = 0; d < D; d++)
				if (st0 + d < st1) issue(d, st0 + d);
			{
				int younger = st1 - 1 - st0;
				sd_wait(younger > D - 1 ? D - 1 : younger);  // stage st0 has landed
			}
			__syncthreads();  // A: the converters may read raw stage st0
			{
				int younger = st1 - 2 - st0;
				if (st0 + 1 < st1) sd_wait(younger > D - 2 ? D - 2 : younger);  // stage st0 + 1 has landed
			}
			__syncthreads();  // B: digit buffer 0 holds stage st0
			for (int ks = st0; ks < st1; ks++) {
				if (ks + D < st1) issue((ks - st0) % D, ks + D);  // the raw buffer of stage ks was converted during k-step ks - 1
				if (ks + 2 < st1) {
					int younger = st1 - 3 - ks;  // stages issued beyond ks + 2
					sd_wait(younger > D - 2 ? D - 2 : younger);  // stage ks + 2 has landed
				}
				__syncthreads();
			}
			continue;
		}
		if (wid >= 4) {
			// ---- converter waves: raw stage in LDS -> digit images in LDS; sums of squares of their 32 rows ----
			const int cw = wid - 4, lr = lane >> 3, c4 = (lane & 7) * 4;
			int sh[4];
			double sq[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int64_t r = row0 + cw * 32 + 8 * j + lr;
				sh[j] = ysh[r < rows ? r : 0];
				sq[j] = 0.0;
			}
			auto convert = [&](int rbuf, int abuf, int ks) {
				if (SQ_EXP & 2) return;
				const char* raw = rlds + rbuf * RSTAGE + (cw * 32 + lr) * 128 + (lane & 7) * 16;
				char* a = alds + abuf * ASTAGE + (cw * NS) * 1024;
				const bool inside = (int64_t)ks * 32 + c4 < n;  // (cells from n to the end of the k-step count as zeros)
#pragma unroll
				for (int j = 0; j < 4; j++) {
					float4 v = *reinterpret_cast<const float4*>(raw + j * 8 * 128);
					if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
					if (inside && (int64_t)ks * 32 + c4 + 4 > n) {  // the row ends inside these four cells
						const int64_t k = (int64_t)ks * 32 + c4;
						if (k + 1 >= n) v.y = 0.f;
						if (k + 2 >= n) v.z = 0.f;
						if (k + 3 >= n) v.w = 0.f;
					}
					double x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
					for (int i = 0; i < 4; i++) sq[j] = fma(x[i], x[i], sq[j]);
					unsigned w[NS];
					nrm_digits4<NS>(x, sh[j], w);
					const int rr = j * 8 + lr;  // row of the 32-row block; halves swapped when (rr >> 3) & 1 = j & 1
					char* dst = a + (2 * rr) * 16 + (((c4 >> 4) ^ (j & 1)) << 4) + (c4 & 15);
#pragma unroll
					for (int q = 0; q < NS; q++) *reinterpret_cast<unsigned*>(dst + q * 1024) = w[q];
				}
			};
			__syncthreads();  // A
			if (st0 < st1) convert(0, 0, st0);
			__syncthreads();  // B
			for (int ks = st0; ks < st1; ks++) {
				if (ks + 1 < st1) convert((ks + 1 - st0) % D, (ks + 1 - st0) & 1, ks + 1);
				__syncthreads();
			}
			double* sbase = slab ? slab + SD_TM * SQN : ss + row0;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				double v = sq[j];
				v += __shfl_xor(v, 1, 64);
				v += __shfl_xor(v, 2, 64);
				v += __shfl_xor(v, 4, 64);
				if ((lane & 7) == 0) sbase[cw * 32 + 8 * j + lr] = v;
			}
			continue;
		}
		// ---- MFMA waves: 32 rows x 32 Z rows each ----
		i16_t acc[NS];
		auto clear = [&]() {
#pragma unroll
			for (int w = 0; w < NS; w++)
#pragma unroll
				for (int j = 0; j < 16; j++) acc[w][j] = 0;
		};
		double* gbase = slab ? slab + (wid * 32) * SQN : G + (row0 + wid * 32) * SQN;
		double* dbase = slab ? slab + SD_TM * (SQN + 1) + (wid * 32) * SQ_DIG : dig + (row0 + wid * 32) * SQ_DIG;
		const int zsh_l = zsh[lane & 31];
		auto flush = [&](bool first) {
#pragma unroll
			for (int q = 0; q < 16; q++) {
				const int rr = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
				long long lo = 0, hi = 0;
#pragma unroll
				for (int w = 0; w < NS; w++) {
					if (w < 3)
						lo += (long long)acc[w][q] << (8 * w);
					else
						hi += (long long)acc[w][q] << (8 * (w - 3));
				}
				double v = fma((double)hi, 16777216.0, (double)lo);
				const int64_t r = row0 + wid * 32 + rr;
				v = ldexp(v, ysh[r < rows ? r : 0] + zsh_l + 8 * (NS - 1));
				double* o = gbase + (int64_t)rr * SQN + (lane & 31);
				*o = first ? v : *o + v;
				if ((lane & 31) == SQN - 1) {
#pragma unroll
					for (int w = 0; w < SQ_DIG; w++) {
						const double dv = w < NS - 1 ? (double)acc[w][q] * 0.03125 : 0.0;
						double* od = dbase + (int64_t)rr * SQ_DIG + w;
						*od = first ? dv : *od + dv;
					}
				}
			}
		};
		clear();
		int in_chunk = 0;
		bool first = true;
		__syncthreads();  // A
		__syncthreads();  // B: digit buffer 0 holds stage st0, its Z planes have landed
		for (int ks = st0; ks < st1; ks++) {
			const char* a = alds + ((ks - st0) & 1) * ASTAGE + (wid * NS) * 1024 + pos;
			const char* z = zlds + ((ks - st0) % ZD) * ZSTAGE + pos;
			i4_t fa[NS], fb[NS];
			if (SQ_EXP & 1) {
				__syncthreads();
				continue;
			}
			fa[0] = *reinterpret_cast<const i4_t*>(a);
			fb[NS - 1] = *reinterpret_cast<const i4_t*>(z + (NS - 1) * 1024);
			fa[1] = *reinterpret_cast<const i4_t*>(a + 1024);
			fb[NS - 2] = *reinterpret_cast<const i4_t*>(z + (NS - 2) * 1024);
#pragma unroll
			for (int sI = 0; sI < NS; sI++) {
#pragma unroll
				for (int tI = NS - 1 - sI; tI < NS; tI++) acc[sI + tI - (NS - 1)] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[sI], fb[tI], acc[sI + tI - (NS - 1)], 0, 0, 0);
				if (sI + 2 < NS) {
					fa[sI + 2] = *reinterpret_cast<const i4_t*>(a + (sI + 2) * 1024);
					fb[NS - 3 - sI] = *reinterpret_cast<const i4_t*>(z + (NS - 3 - sI) * 1024);
				}
			}
			__syncthreads();
			if (++in_chunk >= SQ_FLUSH && ks + 1 < st1) {
				flush(first);
				first = false;
				clear();
				in_chunk = 0;
			}
		}
		if (st0 < st1)
			flush(first);
		else if (slab) {
			clear();
			flush(true);
		}
	}
}

// Sum the slabs of every split row tile in workgroup order (deterministic) into G, ss and dig.
__global__ void __launch_bounds__(256) k_skinny_i8_fixup(double* __restrict__ G, double* __restrict__ ss, double* __restrict__ dig, SkinnyQSched s) {
	const int ts = blockIdx.x;
	const int64_t u0 = (int64_t)ts * s.nkt, u1 = u0 + s.nkt;
	const int first = (int)(u0 / s.units_per_wg);
	int last = (int)((u1 - 1) / s.units_per_wg);
	if (last > s.nwg - 1) last = s.nwg - 1;
	if (first == last && (int64_t)first * s.units_per_wg <= u0 && (int64_t)(first + 1) * s.units_per_wg >= u1) return;  // written whole
	const int64_t t = s.tiles_dp + ts;
	const int e = (blockIdx.y * 256 + threadIdx.x) * 2;
	const int64_t slab = (int64_t)s.tm * (SQN + 1 + SQ_DIG);
	if (e >= slab) return;
	const int first_local = ((int64_t)first * s.units_per_wg / s.nkt) == ts ? 0 : 1;
	const double* src = s.work + ((int64_t)2 * first + first_local) * slab + e;
	d2v_t acc = *reinterpret_cast<const d2v_t*>(src);
	src += (int64_t)(2 - first_local) * slab;
	for (int p = first + 1; p <= last; p++, src += 2 * slab) acc += *reinterpret_cast<const d2v_t*>(src);
	if (e < s.tm * SQN)
		*reinterpret_cast<d2v_t*>(G + t * s.tm * SQN + e) = acc;
	else if (e < s.tm * (SQN + 1))
		*reinterpret_cast<d2v_t*>(ss + t * s.tm + (e - s.tm * SQN)) = acc;
	else
		*reinterpret_cast<d2v_t*>(dig + t * s.tm * SQ_DIG + (e - s.tm * (SQN + 1))) = acc;
}

// Row scales of a resident matrix for the integer engines: ysh[row] = e - 46 with 2^e > max_k |Y[row,k]| (frexp), and the rows'
// sums of squares (the caller compares them with what a later streaming pass finds: the scales are only valid for the same rows).
template <typename T>
__global__ void __launch_bounds__(256) k_row_scales(const T* __restrict__ Y, int64_t rows, int64_t n, int64_t ldy, int* __restrict__ ysh,
													double* __restrict__ ss) {
	const int lane = threadIdx.x & 63;
	const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (row >= rows) return;
	const T* y = Y + row * ldy;
	double mx = 0.0, sq = 0.0;
	for (int64_t k = (int64_t)lane * 4; k < n; k += 256) {
		Row4<T> v;
		v.load(y + k);
		double x[4];
		v.get(x);
#pragma unroll
		for (int i = 0; i < 4; i++) {
			if (k + i < n) {
				mx = fmax(mx, fabs(x[i]));
				sq = fma(x[i], x[i], sq);
			}
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		mx = fmax(mx, __shfl_xor(mx, o, 64));
		sq += __shfl_xor(sq, o, 64);
	}
	if (lane == 0) {
		int e = 0;
		if (mx > 0.0 && mx < INFINITY) (void)frexp(mx, &e);
		ysh[row] = e - (8 * SQ_NS - 2);
		ss[row] = sq;
	}
}

static int g_num_cu_sq = 0;
static int sq_num_cu() {
	if (g_num_cu_sq == 0) {
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&g_num_cu_sq, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || g_num_cu_sq <= 0)
			g_num_cu_sq = 256;
	}
	return g_num_cu_sq;
}

extern "C" int64_t nrm_skinny_i8_workspace_bytes(void) { return (int64_t)2 * sq_num_cu() * SQ_SLAB * (int64_t)sizeof(double); }

extern "C" int nrm_row_scales(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, int32_t* d_ysh, double* d_ss, void* stream) {
	NRM_REQUIRE(y_dtype == NRM_F32 || y_dtype == NRM_F64, "nrm_row_scales: bad dtype");
	NRM_REQUIRE(rows > 0 && n > 0 && ldy >= (n + 3) / 4 * 4 && d_y && d_ysh && d_ss, "nrm_row_scales: bad arguments");
	NRM_REQUIRE(ldy % (16 / (y_dtype == NRM_F64 ? 8 : 4)) == 0 && (uintptr_t)d_y % 16 == 0, "nrm_row_scales: rows must be 16-byte aligned");
	dim3 grid((unsigned)((rows + 3) / 4));
	if (y_dtype == NRM_F64)
		hipLaunchKernelGGL(k_row_scales<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)d_y, rows, n, ldy, d_ysh, d_ss);
	else
		hipLaunchKernelGGL(k_row_scales<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)d_y, rows, n, ldy, d_ysh, d_ss);
	return nrm_check_launch("k_row_scales");
}

extern "C" int nrm_skinny_i8(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, const int32_t* d_ysh, const void* d_zq,
							 const int32_t* d_zsh, int64_t k_pad, double* d_g, double* d_ss, double* d_dig, int64_t rows_pad, void* d_work,
							 void* stream) {
	NRM_REQUIRE(y_dtype == NRM_F32 || y_dtype == NRM_F64, "nrm_skinny_i8: bad dtype");
	NRM_REQUIRE(rows > 0 && n > 0 && n < (1 << 22), "Incorrect dx/dy/dc size.");
	const int64_t n16 = (n + 15) / 16 * 16;
	NRM_REQUIRE(ldy >= n16, "nrm_skinny_i8: rows must be readable (and zero) up to a multiple of 16 cells: ldy >= %lld", (long long)n16);
	NRM_REQUIRE(k_pad >= n && k_pad % SQC == 0, "nrm_skinny_i8: Z must be padded to a multiple of %d cells", SQC);
	NRM_REQUIRE(rows_pad >= rows && rows_pad % SQ_TM == 0, "nrm_skinny_i8: rows_pad must be a multiple of %d", SQ_TM);
	NRM_REQUIRE(d_y && d_ysh && d_zq && d_zsh && d_g && d_ss && d_dig && d_work, "nrm_skinny_i8: null pointer");
	const int64_t al = 16 / (y_dtype == NRM_F64 ? 8 : 4);
	NRM_REQUIRE(ldy % al == 0 && (uintptr_t)d_y % 16 == 0 && (uintptr_t)d_zq % 16 == 0, "nrm_skinny_i8: rows must be 16-byte aligned");
	hipStream_t st = (hipStream_t)stream;
	SkinnyQSched s;
	s.work = (double*)d_work;
	static int variant = -1;  // NRM_DE_I8_DMA=0: fp32 rows through registers too (the first form of this kernel)
	if (variant < 0) {
		const char* e = getenv("NRM_DE_I8_DMA");
		variant = (e && e[0] == '0') ? 0 : 1;
	}
	const bool dma = y_dtype == NRM_F32 && variant == 1;
	s.tm = dma ? SD_TM : SQ_TM;
	const int64_t tiles = rows_pad / s.tm;
	s.nkt = (int)(k_pad / SQC);
	s.nwg = sq_num_cu();
	s.nwg -= s.nwg % 8;
	const int64_t waves = tiles / s.nwg, rem = tiles - waves * s.nwg;
	s.tiles_sk = (int)rem;
	s.tiles_dp = (int)(tiles - rem);
	s.units_per_wg = (int)((rem * s.nkt + s.nwg - 1) / s.nwg);
	const int64_t zplane = (k_pad / 32) * 1024;  // one 32-row block: a plane is ceil(k_pad / 32) KB images
	if (y_dtype == NRM_F64)
		hipLaunchKernelGGL(k_skinny_i8<double>, dim3((unsigned)s.nwg), dim3(512), 0, st, (const double*)d_y, rows, n, ldy, d_ysh, (const char*)d_zq, zplane,
						   d_zsh, d_g, d_ss, d_dig, s);
	else if (dma)
		hipLaunchKernelGGL(k_skinny_i8_dma, dim3((unsigned)s.nwg), dim3(768), 0, st, (const float*)d_y, rows, n, ldy, d_ysh, (const char*)d_zq, zplane, d_zsh,
						   d_g, d_ss, d_dig, s);
	else
		hipLaunchKernelGGL(k_skinny_i8<float>, dim3((unsigned)s.nwg), dim3(512), 0, st, (const float*)d_y, rows, n, ldy, d_ysh, (const char*)d_zq, zplane,
						   d_zsh, d_g, d_ss, d_dig, s);
	if (s.tiles_sk > 0)
		hipLaunchKernelGGL(k_skinny_i8_fixup, dim3((unsigned)s.tiles_sk, (unsigned)((s.tm * (SQN + 1 + SQ_DIG) / 2 + 255) / 256)), dim3(256), 0, st, d_g, d_ss, d_dig, s);
	return nrm_check_launch("k_skinny_i8");
}
