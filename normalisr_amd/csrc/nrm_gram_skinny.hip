// K2s: de with few design rows (case-control DE, config C3: 1 x 20k x 100k cells) is HBM-bound -- every
// expression value is needed once.  This kernel streams the RAW expression rows (fp32 or fp64, as handed
// over by numpy, never materialised as fp64 residuals) exactly once and contracts them on the fp64 matrix
// cores against Z = [C; X~] (covariates and the already-residualised design rows, <= 32 rows, fp64):
//
//     G[y, :] = sum_k Y[y,k] Z[:,k]        (ny, 32)      ss[y] = sum_k Y[y,k]^2
//
// from which the per-pair sweep (k_de_small_sweep) recovers, without ever forming y~:
//     y~ . x~ = y . x~                      (x~ is orthogonal to C)           association.py:234
//     |y~|^2  = |y|^2 - (y C^T) dci (C y^T)                                   association.py:229-230
// Algorithmic HBM bytes: itemsize * n per expression row (+ Z from L2).
//
// Structure (round 2; the round-1 kernel loaded the expression rows into registers from the MFMA waves themselves and
// re-staged Z with load + ds_write: 2.35 ms on C3, the loads and the matrix cores adding up instead of overlapping):
//  * one workgroup per CU owns 256 rows: 8 (or 4) MFMA waves of 32 (64) rows and 4 LOADER waves that do nothing but
//    issue global -> LDS DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write) -- a wave blocked on a full
//    memory queue cannot feed the matrix cores, so the two jobs live in different waves (12 waves = 3 per SIMD);
//  * a stage is 128 bytes (one line) of each of the 256 rows plus the matching cells of Z, 4 rows x 256 B / 8 rows x 128 B per
//    wave instruction; the ring holds D = 4 stages (128 KB of rows + <= 32 KB of Z), three of them in flight or
//    landed ahead of the one being contracted;
//  * the DMA is issued from inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of the first ds_read after every
//    LDS-DMA it can see (it cannot prove that the read does not alias the DMA's destination), which serialises a wave's
//    loads with its own work -- measured 2.40 ms with the builtin where loads alone took 1.68 and the MFMA loop alone
//    1.62 ms.  Waits are counted by hand: a loader waits until all but the instructions of the D - 2 younger stages have
//    landed (s_waitcnt vmcnt(N)), then ONE workgroup barrier per stage publishes stage s to the MFMA waves and tells the
//    loaders that stage s - 1 has been read, whose buffer is refilled at once with stage s + D - 1;
//  * LDS images are read back with conflict-free ds_read_b128: 16-byte chunk c of row R sits at position
//    c ^ (R & (chunks per row - 1)) -- applied to the SOURCE chunk each lane fetches (the DMA writes LDS in lane order);
//    the operands of slab s + 1 are read while the MFMAs of slab s run (two register sets);
//  * MFMA operand maps as before: lane (r = l & 15, g = l >> 4) feeds cells 4g..4g+3 of row r to MFMA steps 0..3 (a dot
//    product does not care about the order of its terms as long as Z uses the same permutation); NT 16-row Z tiles on
//    v_mfma_f64_16x16x4 and NQ 4-row groups on v_mfma_f64_4x4x4 (17-24 Z rows cost 64 + 16 NQ cycles per step, not 128).
// Measured on C3 (20 000 x 100 000 fp32, 21 Z rows; tools/k2s_exp.sh, parts compiled out with -DSK_EXP): loaders alone
// 1.33 ms (6.0 TB/s), MFMA waves alone 1.73 ms, both 2.03 ms; with every load served from L2 1.81 ms -- the last 0.2 ms
// appear only with HBM traffic under the fp64 matrix cores.  16 Z rows: 1.65 ms, 32: 2.33 ms (round 1: 1.76 / 2.35 / 2.74).
// Persistent DP + stream-K schedule as in K2 so that 79 row tiles still fill 256 CUs; partial pieces go to
// workspace slabs and are summed in a fixed order (k_skinny_fixup): bitwise reproducible, no atomics.
#include "nrm_common.h"

#define SKN 32       // columns of G
#define SKC 128      // cells per schedule unit (stream-K pieces are cut at multiples of it; Z is padded to it)

typedef double d2_t __attribute__((ext_vector_type(2)));

struct SkinnySched {
	int nkt, tiles_dp, tiles_sk, units_per_wg, nwg;
	int tm;        // rows per workgroup tile
	double cval;   // != 0: column 31 of G receives cval * sum_k Y[y,k] -- the product with a CONSTANT Z row (the intercept)
	               // taken out of Z and summed on the vector ALU (one add per value) instead of a 4-row matrix-core group
	double* work;  // per partial piece: a (tm x 32) slab of G followed by tm sums of squares; two pieces per workgroup
};

#ifndef SK2_TM
#define SK2_TM 256  // rows per workgroup tile
#endif
#ifndef SK2_D
#define SK2_D 4  // ring depth (stages)
#endif
#ifndef SK2_RT
#define SK2_RT 4  // 16-row tiles per MFMA wave (4: four waves of 64 rows; 2: eight waves of 32 rows)
#endif
#ifndef SK2_NL
#define SK2_NL 4  // loader waves (vmcnt counts at most 63 outstanding instructions per wave: a 256-row stage is 32 + NZI of them)
#endif
#ifndef SK_EXP
#define SK_EXP 0  // timing experiments (tools/build_exp.sh): 1 no MFMA, 2 no DMA, 4 no sum of squares, 8 no barrier
#endif

template <typename T>
struct LdsSlab;  // the 4 cells a lane feeds to the MFMA steps of one slab, read from the swizzled LDS image
template <>
struct LdsSlab<float> {
	float4 v;
	__device__ __forceinline__ void load(const char* row, int sl, int lg, int swz) {
		v = *reinterpret_cast<const float4*>(row + (((4 * sl + lg) ^ swz) << 4));
	}
	__device__ __forceinline__ void fake(int lg) { v = make_float4(1.f + lg, 2.f, 3.f, 4.f); }
	__device__ __forceinline__ double get(int s) const { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }
};
template <>
struct LdsSlab<double> {
	d2_t a, b;
	__device__ __forceinline__ void load(const char* row, int sl, int lg, int swz) {
		const int pos = (8 * sl + 2 * lg) ^ swz;
		a = *reinterpret_cast<const d2_t*>(row + (pos << 4));
		b = *reinterpret_cast<const d2_t*>(row + ((pos ^ 1) << 4));
	}
	__device__ __forceinline__ void fake(int lg) {
		a = (d2_t){1.0 + lg, 2.0};
		b = (d2_t){3.0, 4.0};
	}
	__device__ __forceinline__ double get(int s) const { return s == 0 ? a[0] : s == 1 ? a[1] : s == 2 ? b[0] : b[1]; }
};

// One LDS-DMA instruction the compiler does not see (M0 = wave-uniform LDS byte address; cdna_hip_programming.md 5.7): 64 lanes x 16 B.
__device__ __forceinline__ void sk_dma16(const void* gsrc, unsigned lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
				 : "=&s"(keep)
				 : "v"(gsrc), "s"(lds_dst)
				 : "memory");
}
template <int PS>
__device__ __forceinline__ void sk_wait_stages(int younger) {  // all DMA instructions done except those of `younger` stages (PS each)
	if (younger <= 0)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	else if (younger == 1)
		asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PS) : "memory");
	else
		asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PS) : "memory");
}

template <typename T, int NT, int NQ, int RT, int D, int TM, int NL>
__global__ void __launch_bounds__(TM * 4 / RT + 64 * NL) k_gram_skinny(const T* __restrict__ A, int64_t rows, int64_t n16, int64_t lda,
																  const double* __restrict__ Z, int64_t ldz, double* __restrict__ G,
																  double* __restrict__ ss, SkinnySched s) {
	constexpr int NW = TM / (16 * RT);            // MFMA waves per workgroup, each owning RT x 16 rows; wave NW is the loader
	constexpr int SB = (D == 2 ? 256 : 128);      // bytes of a row per stage
	constexpr int SC = SB / (int)sizeof(T);       // cells per stage
	constexpr int NSL = SC / 16;                  // slabs (16 cells) per stage
	constexpr int YCH = SB / 16;                  // 16-byte chunks of an expression row per stage
	constexpr int YRPI = 64 / YCH;                // expression rows per DMA instruction
	constexpr int NY = TM / YRPI;                 // expression DMA instructions per stage
	constexpr int ZR = NT * 16 + NQ * 4;          // Z rows used
	constexpr int ZROWB = SC * 8;                 // bytes of one Z row per stage
	constexpr int ZCH = ZROWB / 16;               // its 16-byte chunks
	constexpr int ZRPI = 64 / ZCH;                // Z rows per DMA instruction
	constexpr int NZI = (ZR + ZRPI - 1) / ZRPI;   // Z DMA instructions per stage (whole instructions: up to ZRPI - 1 unused rows of the 32-row Z buffer come along)
	constexpr int ZSW = (ZCH < 16 ? ZCH : 16) - 1;  // swizzle mask of a Z row
	constexpr int SPU = SKC / SC;                 // stages per schedule unit (SKC cells)
	constexpr int EPC = 16 / (int)sizeof(T);      // elements per 16-byte chunk
	constexpr int YSTAGE = TM * SB, ZSTAGE = NZI * 1024;
	constexpr int NYL = NY / NL;                  // expression DMA instructions per loader wave and stage
	constexpr int CZ0 = NZI / NL, NZHI = NZI % NL;  // Z instructions are dealt round-robin: loaders < NZHI issue CZ0 + 1, the others CZ0
	static_assert(NSL >= 1 && NZI * ZRPI <= 32 && ZCH <= 32 && D >= 2 && D <= 4 && SB * D <= 512, "unsupported stage geometry");
	static_assert(NY % NL == 0 && 2 * (NYL + CZ0 + 1) < 64, "loader split / vmcnt field");
	__shared__ __attribute__((aligned(1024))) char ylds[D * YSTAGE];
	__shared__ __attribute__((aligned(1024))) char zlds[D * ZSTAGE];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l15 = lane & 15, lg = lane >> 4;
	const int per_xcd = s.nwg >> 3;
	const int p = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
	int t_dp = p;
	int64_t u = (int64_t)p * s.units_per_wg;
	const int64_t total = (int64_t)s.tiles_sk * s.nkt;
	int64_t uend = u + s.units_per_wg;
	if (uend > total) uend = total;
	int sk_piece = 0;
	typedef __attribute__((address_space(3))) char* lds_ptr_t;
	const unsigned ylds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)ylds);
	const unsigned zlds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)zlds);
	for (;;) {
		int t, c0, c1;  // tile, unit range [c0, c1) in units of SKC cells
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
			if (!(c0 == 0 && c1 == s.nkt)) slab = s.work + ((int64_t)2 * p + sk_piece) * (TM * SKN + TM);
			sk_piece++;
		} else {
			break;
		}
		const int st0 = c0 * SPU;
		int st1 = c1 * SPU;  // stages [st0, st1), cut at the end of the (16-padded) rows
		{
			const int last = (int)((n16 + SC - 1) / SC);
			if (st1 > last) st1 = last;
		}
		__syncthreads();  // previous piece: every MFMA wave is done with LDS
		if (wid >= NW) {
			// ---- loader waves: nothing but DMA issue, counted waits and the stage barrier ----
			const int li = wid - NW;
			// source pointers (cell 0 of the row + the swizzled chunk); rows past the end are clamped to row 0: their
			// products land in padding rows of G / ss that nobody reads.  Instruction j covers rows YRPI*j.., lane -> (row, position)
			const int jr = lane / YCH, jp = lane % YCH;
			const int zr_in = lane / ZCH, zp = lane % ZCH;
			const T* ysrc[NYL];
#pragma unroll
			for (int j = 0; j < NYL; j++) {
				const int rr = YRPI * (li + j * NL) + jr;
				const int64_t r = (int64_t)t * TM + rr;
				ysrc[j] = A + (r < rows ? r : 0) * lda + (jp ^ (rr & (YCH - 1))) * EPC;
			}
			int ycell[NYL];  // first cell (within a stage) of the chunk this lane fetches with instruction j
#pragma unroll
			for (int j = 0; j < NYL; j++) ycell[j] = (jp ^ ((YRPI * (li + j * NL) + jr) & (YCH - 1))) * EPC;
			const double* zsrc[CZ0 + 1];
#pragma unroll
			for (int q = 0; q < CZ0 + 1; q++) {
				const int zr = (li + q * NL) * ZRPI + zr_in;  // (the last one is only issued by loaders < NZHI)
				zsrc[q] = Z + (int64_t)(zr < 32 ? zr : 0) * ldz + ((zp ^ (zr & ZSW)) << 1);
			}
			const bool zhi = li < NZHI;
			auto issue = [&](int buf, int64_t k0) {
				if (SK_EXP & 2) return;
				if (SK_EXP & 64) k0 &= 127;  // experiment: every load hits L2 (the first 512 bytes of each row over and over)
#pragma unroll
				for (int q = 0; q < CZ0; q++) sk_dma16(zsrc[q] + k0, zlds0 + buf * ZSTAGE + (li + q * NL) * 1024);
				if (zhi) sk_dma16(zsrc[CZ0] + k0, zlds0 + buf * ZSTAGE + (li + CZ0 * NL) * 1024);
				const bool tail = k0 + SC > n16;  // a chunk past the (16-padded) row end: fetch anything valid, its slab is skipped
#pragma unroll
				for (int j = 0; j < NYL; j++) sk_dma16((tail && k0 + ycell[j] >= n16) ? A : ysrc[j] + k0, ylds0 + buf * YSTAGE + (li + j * NL) * 1024);
			};
			for (int d = 0; d < D - 1; d++)
				if (st0 + d < st1) issue(d, (int64_t)(st0 + d) * SC);
			for (int sg = st0; sg < st1; sg++) {
				int younger = st1 - 1 - sg;  // stages issued beyond this one
				if (younger > D - 2) younger = D - 2;
				if (zhi)  // this loader's share of stage sg has landed
					sk_wait_stages<NYL + CZ0 + 1>(younger);
				else
					sk_wait_stages<NYL + CZ0>(younger);
				if (!(SK_EXP & 8)) __syncthreads();  // it is visible to the MFMA waves, and they are done with stage sg - 1 ...
				if (sg + D - 1 < st1) issue((sg - st0 + D - 1) % D, (int64_t)(sg + D - 1) * SC);  // ... whose buffer is refilled at once
			}
			continue;
		}
		// ---- MFMA waves: LDS reads and matrix cores only ----
		d4_t acc[RT][NT];
#pragma unroll
		for (int i = 0; i < RT; i++)
#pragma unroll
			for (int j = 0; j < NT; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
		double accq[RT][NQ ? NQ : 1];
#pragma unroll
		for (int i = 0; i < RT; i++)
#pragma unroll
			for (int g = 0; g < (NQ ? NQ : 1); g++) accq[i][g] = 0.0;
		double sq[RT], sy[RT];
#pragma unroll
		for (int i = 0; i < RT; i++) sq[i] = sy[i] = 0.0;
		const bool sumcol = s.cval != 0.0;
		struct Ops {
			LdsSlab<T> cur[RT];
			double zf[NT][4];
			double zq[NQ ? NQ : 1][4];
		};
		auto load_ops = [&](Ops& o, int buf, int sl) {
			const char* yrow = ylds + buf * YSTAGE + (wid * (RT * 16) + l15) * SB;
			const char* zrow = zlds + buf * ZSTAGE + l15 * ZROWB;
			const char* zqrow = zlds + buf * ZSTAGE + (NT * 16 + (l15 & 3)) * ZROWB;
			const int yswz = l15 & (YCH - 1);
#pragma unroll
			for (int i = 0; i < RT; i++) {
				if (SK_EXP & 16)
					o.cur[i].fake(lg);
				else
					o.cur[i].load(yrow + i * 16 * SB, sl, lg, yswz);
			}
			const int zc = 8 * sl + 2 * lg;
#pragma unroll
			for (int j = 0; j < NT; j++) {
				const int pos = zc ^ (l15 & ZSW);
				const d2_t z01 = *reinterpret_cast<const d2_t*>(zrow + j * 16 * ZROWB + (pos << 4));
				const d2_t z23 = *reinterpret_cast<const d2_t*>(zrow + j * 16 * ZROWB + ((pos ^ 1) << 4));
				o.zf[j][0] = z01[0];
				o.zf[j][1] = z01[1];
				o.zf[j][2] = z23[0];
				o.zf[j][3] = z23[1];
			}
#pragma unroll
			for (int g = 0; g < NQ; g++) {
				const int pos = zc ^ ((4 * g + (l15 & 3)) & ZSW);
				const d2_t z01 = *reinterpret_cast<const d2_t*>(zqrow + g * 4 * ZROWB + (pos << 4));
				const d2_t z23 = *reinterpret_cast<const d2_t*>(zqrow + g * 4 * ZROWB + ((pos ^ 1) << 4));
				o.zq[g][0] = z01[0];
				o.zq[g][1] = z01[1];
				o.zq[g][2] = z23[0];
				o.zq[g][3] = z23[1];
			}
		};
		auto contract = [&](const Ops& o) {
#pragma unroll
			for (int st = 0; st < 4; st++)
#pragma unroll
				for (int i = 0; i < RT; i++) {
					double a = o.cur[i].get(st);
					if (SK_EXP & 32) a = __hiloint2double(__float_as_int((float)a), 0x3ff00000);  // experiment: no v_cvt_f64_f32
					if (!(SK_EXP & 4)) sq[i] = fma(a, a, sq[i]);
					if (sumcol) sy[i] += a;
					if (SK_EXP & 1) {
						acc[i][0][st] += a;  // keeps the operand reads alive without the matrix cores
					} else {
#pragma unroll
						for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, o.zf[j][st], acc[i][j], 0, 0, 0);
#pragma unroll
						for (int g = 0; g < NQ; g++) accq[i][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, o.zq[g][st], accq[i][g], 0, 0, 0);
					}
				}
		};
		for (int sg = st0; sg < st1; sg++) {
			const int buf = (sg - st0) % D;
			const int64_t k0 = (int64_t)sg * SC;
			if (!(SK_EXP & 8)) __syncthreads();  // stage sg has landed (the loader waited for it before this barrier)
			int nsl = (int)((n16 - k0) / 16);
			if (nsl > NSL) nsl = NSL;
			// operands of slab sl + 1 are read from LDS while the MFMAs of slab sl run (two register sets)
			Ops ops[2];
			load_ops(ops[0], buf, 0);
#pragma unroll
			for (int sl = 0; sl < NSL; sl++) {
				if (sl + 1 < NSL) load_ops(ops[(sl + 1) & 1], buf, sl + 1);
				if (sl < nsl) contract(ops[sl & 1]);
			}
		}
		// a piece covering all cells writes G / ss directly; partial pieces write their slab, summed in fixed order by k_skinny_fixup
		double* gbase = slab ? slab + (wid * (RT * 16)) * SKN : G + ((int64_t)t * TM + wid * (RT * 16)) * SKN;
		double* sbase = slab ? slab + TM * SKN + wid * (RT * 16) : ss + (int64_t)t * TM + wid * (RT * 16);
#pragma unroll
		for (int i = 0; i < RT; i++)
#pragma unroll
			for (int j = 0; j < NT; j++)
#pragma unroll
				for (int q = 0; q < 4; q++) gbase[(int64_t)(i * 16 + lg + 4 * q) * SKN + j * 16 + l15] = acc[i][j][q];
		if (NT < 2) {  // columns not covered by a 16-row tile: the 4-row groups, zeros beyond them
			const int yrow2 = 4 * ((lane >> 2) & 3) + lg, zc2 = lane & 3;
#pragma unroll
			for (int i = 0; i < RT; i++)
#pragma unroll
				for (int g = 0; g < 4; g++) {
					double v = 0.0;
#pragma unroll
					for (int gg = 0; gg < NQ; gg++)
						if (gg == g) v = accq[i][gg];
					gbase[(int64_t)(i * 16 + yrow2) * SKN + 16 + 4 * g + zc2] = v;
				}
		}
#pragma unroll
		for (int i = 0; i < RT; i++) {
			double v = sq[i];
			v += __shfl_xor(v, 16, 64);
			v += __shfl_xor(v, 32, 64);
			if (lg == 0) sbase[i * 16 + l15] = v;
			if (sumcol) {  // (after the column stores above: the last column is this wave's to overwrite)
				double w = sy[i];
				w += __shfl_xor(w, 16, 64);
				w += __shfl_xor(w, 32, 64);
				if (lg == 0) gbase[(int64_t)(i * 16 + l15) * SKN + (SKN - 1)] = s.cval * w;
			}
		}
	}
}

// Sum the slabs of every split row tile in workgroup order (deterministic) into G and ss.  blockIdx.y selects a
// 256-element chunk of the slab so that enough loads are in flight.
__global__ void __launch_bounds__(256) k_skinny_fixup(double* __restrict__ G, double* __restrict__ ss, SkinnySched s) {
	const int ts = blockIdx.x;
	const int64_t u0 = (int64_t)ts * s.nkt, u1 = u0 + s.nkt;
	const int first = (int)(u0 / s.units_per_wg);
	int last = (int)((u1 - 1) / s.units_per_wg);
	if (last > s.nwg - 1) last = s.nwg - 1;
	if (first == last && (int64_t)first * s.units_per_wg <= u0 && (int64_t)(first + 1) * s.units_per_wg >= u1) return;  // written whole
	const int64_t t = s.tiles_dp + ts;
	const int e = (blockIdx.y * 256 + threadIdx.x) * 2;  // two consecutive elements per thread (16-byte loads)
	const int64_t slab_sz = (int64_t)s.tm * SKN + s.tm;
	if (e >= slab_sz) return;
	// only the first contributing workgroup can have started in the previous tile (then this tile holds its second piece)
	const int first_local = ((int64_t)first * s.units_per_wg / s.nkt) == ts ? 0 : 1;
	const double* src = s.work + ((int64_t)2 * first + first_local) * slab_sz + e;
	d2_t acc = *reinterpret_cast<const d2_t*>(src);
	src += (int64_t)(2 - first_local) * slab_sz;  // first piece of workgroup first + 1
	for (int p = first + 1; p <= last; p++, src += 2 * slab_sz) acc += *reinterpret_cast<const d2_t*>(src);
	if (e < s.tm * SKN)
		*reinterpret_cast<d2_t*>(G + t * s.tm * SKN + e) = acc;
	else
		*reinterpret_cast<d2_t*>(ss + t * s.tm + (e - s.tm * SKN)) = acc;
}

static int g_num_cu_s = 0;

extern "C" int64_t nrm_gram_skinny_workspace_bytes(void) {
	if (g_num_cu_s == 0) {
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&g_num_cu_s, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || g_num_cu_s <= 0)
			g_num_cu_s = 256;
	}
	return (int64_t)2 * g_num_cu_s * (SK2_TM * SKN + SK2_TM) * (int64_t)sizeof(double);
}

extern "C" int nrm_gram_skinny(const void* d_a, int a_dtype, int64_t rows, int64_t n, int64_t lda, const double* d_z, int64_t ldz,
							   int64_t k_pad, double* d_g, double* d_ss, int64_t rows_pad, int64_t nz, double const_row_value, void* d_work,
							   void* stream) {
	NRM_REQUIRE(a_dtype == NRM_F32 || a_dtype == NRM_F64, "nrm_gram_skinny: bad dtype");
	NRM_REQUIRE(rows > 0 && n > 0 && lda >= n, "Incorrect dx/dy/dc size.");
	const int64_t n16 = (n + 15) / 16 * 16;
	NRM_REQUIRE(lda >= n16, "nrm_gram_skinny: rows must be readable (and zero) up to a multiple of 16 cells: lda >= %lld", (long long)n16);
	NRM_REQUIRE(k_pad >= n && k_pad % SKC == 0 && ldz >= k_pad && ldz % 2 == 0, "nrm_gram_skinny: Z must be padded to a multiple of %d cells", SKC);
	NRM_REQUIRE(rows_pad >= rows && rows_pad % SK2_TM == 0, "nrm_gram_skinny: rows_pad must be a multiple of %d", SK2_TM);
	NRM_REQUIRE(d_a && d_z && d_g && d_ss, "nrm_gram_skinny: null pointer");
	const int64_t al = 16 / (a_dtype == NRM_F64 ? 8 : 4);
	NRM_REQUIRE(lda % al == 0 && (uintptr_t)d_a % 16 == 0 && (uintptr_t)d_z % 16 == 0, "nrm_gram_skinny: rows must be 16-byte aligned");
	hipStream_t st = (hipStream_t)stream;
	if (g_num_cu_s == 0) {
		int dev = 0;
		NRM_HIP(hipGetDevice(&dev));
		NRM_HIP(hipDeviceGetAttribute(&g_num_cu_s, hipDeviceAttributeMultiprocessorCount, dev));
		if (g_num_cu_s <= 0) g_num_cu_s = 256;
	}
	NRM_REQUIRE(d_work != nullptr, "nrm_gram_skinny: workspace of nrm_gram_skinny_workspace_bytes() bytes required");
	SkinnySched s;
	s.work = (double*)d_work;
	s.tm = SK2_TM;
	s.cval = const_row_value;
	NRM_REQUIRE(const_row_value == 0.0 || nz < SKN, "nrm_gram_skinny: a constant row needs the last column of G free (nz <= 31)");
	const int64_t tiles = rows_pad / s.tm;
	s.nkt = (int)(k_pad / SKC);
	s.nwg = g_num_cu_s;  // one workgroup per CU (its LDS ring takes 130-160 KB)
	s.nwg -= s.nwg % 8;
	const int64_t waves = tiles / s.nwg, rem = tiles - waves * s.nwg;
	const int64_t sk = rem;  // < nwg, so a workgroup's unit range spans at most two tiles (two slabs per workgroup)
	s.tiles_sk = (int)sk;
	s.tiles_dp = (int)(tiles - sk);
	s.units_per_wg = (int)((sk * s.nkt + s.nwg - 1) / s.nwg);
	// nz <= 16 used Z rows: one MFMA column tile instead of two (half the matrix-core work: the pass becomes HBM-bound);
	// 17..24: one column tile plus one or two 4-row groups on the 4x4x4 instruction (80 / 96 cycles per step instead of 128)
	const int variant = (nz > 0 && nz <= 16) ? 0 : (nz > 16 && nz <= 20) ? 1 : (nz > 20 && nz <= 24) ? 2 : 3;
#define SK_LAUNCH(TT, NT_, NQ_)                                                                                                              \
	hipLaunchKernelGGL((k_gram_skinny<TT, NT_, NQ_, SK2_RT, SK2_D, SK2_TM, SK2_NL>), dim3((unsigned)s.nwg), dim3(SK2_TM * 4 / SK2_RT + 64 * SK2_NL), \
					   0, st, (const TT*)d_a, rows, n16, lda, d_z, ldz, d_g, d_ss, s)
	if (a_dtype == NRM_F64) {
		if (variant == 0) SK_LAUNCH(double, 1, 0);
		else if (variant == 1) SK_LAUNCH(double, 1, 1);
		else if (variant == 2) SK_LAUNCH(double, 1, 2);
		else SK_LAUNCH(double, 2, 0);
	} else {
		if (variant == 0) SK_LAUNCH(float, 1, 0);
		else if (variant == 1) SK_LAUNCH(float, 1, 1);
		else if (variant == 2) SK_LAUNCH(float, 1, 2);
		else SK_LAUNCH(float, 2, 0);
	}
#undef SK_LAUNCH
	if (s.tiles_sk > 0)
		hipLaunchKernelGGL(k_skinny_fixup, dim3((unsigned)s.tiles_sk, (unsigned)(((s.tm * SKN + s.tm) / 2 + 255) / 256)), dim3(256), 0, st, d_g, d_ss, s);
	return nrm_check_launch("k_gram_skinny");
}
