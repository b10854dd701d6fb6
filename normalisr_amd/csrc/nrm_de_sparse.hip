// de with a SPARSE design matrix (gRNA incidence of a CRISPR screen: BASELINE configs[3] has 1000 gRNAs x 50 000 cells with 1 % of the
// entries set).  The reference residualises design and expression rows and multiplies them densely (association.py:224-235); since the
// residual y~ is orthogonal to the covariates,
//     y~ . x~ = y . x - (y C^T) . b_x,        |y~|^2 = |y|^2 - (y C^T) . b_y,       b = (row C^T) (C C^T)^+   (association.py:224-229)
// so all the kernel needs of an expression row are its sums over the few cells where each design row is not zero, its products with the
// covariates and its sum of squares -- ONE pass over the raw expression matrix (HBM) instead of K1's two sweeps and digit planes plus
// K2's 2 n flop per pair: 2 n nx ny flop become nnz(x) ny additions.
//
// A workgroup takes R = 16 / sizeof(T) expression rows (4 fp32, 2 fp64) and walks the cells in chunks of DS_CH: a chunk of its rows lies in
// LDS, cell by cell (the R values of a cell are one 16-byte record), thread t owns the design rows ("slots") t, t + 512, ... (DS_G of them,
// sums in registers) and gathers, for each, the records of the cells where that row is not zero -- one ds_read_b128 per entry serves R
// pairs.  The entries come in ELL form per (chunk, 64 slots of a wave): entries 8 j .. 8 j + 7 of the 64 lanes side by side (one coalesced
// 1 KB load gives every lane its next 8), lists padded to the longest of the 64 (a multiple of 8) with the offset of a record of zeros.
// The next chunk's HBM loads are in flight (in registers) while the current one is gathered.
//
// Round 5: the products with the covariates a = y C^T and |y|^2 are taken INSIDE this kernel, on the fp64 matrix cores, from the chunk of
// records already in LDS (NG >= 0) -- one pass over the expression rows instead of two (rounds 3-4 ran nrm_single1.hip's stream kernel first:
// a second read of the matrix, 0.64 ms of the 2.6 ms step at configs[3] size; still the path for more than DS_FUSED_NC covariates).  Three
// ways to take them on the vector ALU had failed in round 4 (covariate values fetched between the barriers of a chunk: +1.1 ms; a dense phase
// after the gathers: +2.0 ms; values travelling with the rows: +1.0 ms -- 20 doubles per thread and chunk through the registers, 16 x 5 fp64
// FMAs per thread).  v_mfma_f64_4x4x4_4b_f64 does four independent 4 x 4 x 4 products per instruction: block b = the cells = b mod 4 of a group
// of 16, A = 4 covariates x 4 cells, B = 4 cells x the workgroup's 4 rows -- 256 multiply-adds, fully used, 16 cycles of a matrix pipe that is
// otherwise idle here.  Its lane maps (tools/mfma444_probe.hip, measured on gfx950: A[b][i][k] at lane i + 4 b + 16 k, B[b][k][j] at lane
// j + 4 b + 16 k, D[b][i][j] at lane j + 4 b + 16 i) make both operands LANE-LINEAR reads: lane l takes row l & 3 of cell l >> 2 of the group --
// for B the float at byte 4 l of the group's 16 records (one conflict-free ds_read_b32), for A the double at index l of the group's block
// of the covariates stored cell by cell in fours (d_ct: [group of 4 covariates][cell][4], written by k_ds_ct; one coalesced 512-byte load from
// L2).  |y|^2 and the plain sum (a constant covariate: the intercept needs no operand at all) are two vector instructions per lane and cell
// group on the same value.  A wave takes 16 of a chunk's 128 cell groups: 16 loads, 16 LDS reads, 16 conversions, 16 matrix instructions per
// group of 4 covariates.  The fp64 matrix rate of this chip IS its fp64 vector rate -- the matrix instructions take their 16 cycles from the
// units the gathers' conversions and additions run on: what they save is instruction issue and registers, not arithmetic time.
#include "nrm_common.h"
#include "nrm_design.h"  // DS_CH (cells per chunk), DS_T (threads per workgroup), DS_G (design rows per thread)

#define DS_NCMAX 32  // (beyond DS_FUSED_NC the sums over the covariates come from nrm_single1_stream: its limit)
#ifndef DS_UB
#define DS_UB 4  // cell groups per batch of covariate operands (see the matrix-core section of k_de_sparse)
#endif
#ifndef DS_SNAKE
#define DS_SNAKE 1
#endif
#ifndef DS_MFMA_AFTER
#define DS_MFMA_AFTER 0  // the matrix-core section after the gathers of a chunk (1) or before them (0)
#endif
#ifndef DS_NT
#define DS_NT 1  // the expression rows loaded non-temporally when the covariate operands are read beside them (those then stay in L2: 2.00 -> 1.93 ms)
#endif
#define DS_FUSED_NC 8  // covariates besides a constant one whose products with the rows are taken inside k_de_sparse (two groups of 4)
#define DS_WGS 6     // waves per SIMD the register budget is set for (68 registers; three workgroups of 512 threads per CU; 8 waves per SIMD spill and gain nothing)

namespace {

template <typename T>
struct DsVec;
template <>
struct DsVec<float> {
	static constexpr int V = 4, R = 4;  // (8 rows -- 32-byte records, half the index loads per pair, two workgroups per CU: 2.00 ms against 1.56)
};
template <>
struct DsVec<double> {
	static constexpr int V = 2, R = 2;
};

// NG >= 0: the rows' products with the covariates are taken here (NG groups of 4 covariates from ct, plus a constant one when cidx >= 0) and
// written to `common` for later passes; NG = -1: they are read from `common` (nrm_single1_stream, or an earlier pass)
template <typename T, bool ALIGNED, bool BINARY, int NG>
__global__ void __launch_bounds__(DS_T, DS_WGS) k_de_sparse(const T* __restrict__ Y, int64_t ldy, int64_t n, int64_t ny, double* __restrict__ common, int nc,
															   const double* __restrict__ dci, const int16_t* __restrict__ ell, const double* __restrict__ ellv,
															   const int64_t* __restrict__ ellbase, const int32_t* __restrict__ ellw, const int32_t* __restrict__ sig, int ngroups, int group0,
															   const int32_t* __restrict__ slot2x, const double* __restrict__ bx, int64_t ldb,
															   double* __restrict__ dot, int64_t ldd, int by_gene, double* __restrict__ ssy, double* __restrict__ coefy, int32_t* __restrict__ flags, int first,
															   const double* __restrict__ ct, int64_t ctstride, int cidx, double cval) {
	constexpr int V = DsVec<T>::V, R = DsVec<T>::R;
	constexpr int NJ = DS_CH / (DS_T * V);  // groups of V consecutive cells a thread stages per chunk
	typedef T rec_t __attribute__((ext_vector_type(R)));
	typedef T vec_t __attribute__((ext_vector_type(V)));
	typedef short ix_t __attribute__((ext_vector_type(8)));
	__shared__ rec_t lds[DS_CH + 1];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int64_t y0 = (int64_t)blockIdx.x * R;
	const T* row[R];
#pragma unroll
	for (int r = 0; r < R; r++) row[r] = Y + (y0 + r < ny ? y0 + r : ny - 1) * ldy;
	double S[DS_G][R];
#pragma unroll
	for (int r = 0; r < R; r++)
#pragma unroll
		for (int g = 0; g < DS_G; g++) S[g][r] = 0.0;
	if (tid == 0) {
		rec_t z;
#pragma unroll
		for (int r = 0; r < R; r++) z[r] = (T)0;
		lds[DS_CH] = z;  // the record padding entries point at
	}
	const int nchunks = (int)((n + DS_CH - 1) / DS_CH);
	T st[NJ][R][V];  // the next chunk on its way from HBM
	// Every load is issued whatever the cell (no branch that two kinds of loads would meet behind: the compiler waits for loads at such
	// a meeting, and these must stay in flight through the gathers): cells past n read cell 0 instead and count as zeros when used.
	// ALIGNED (the launcher: 16-byte aligned rows and n % V == 0) means a group of V cells is inside or outside as a whole.
	auto request = [&](int c) {
		const int64_t k0 = (int64_t)c * DS_CH;
#pragma unroll
		for (int j = 0; j < NJ; j++) {
			const int64_t k = k0 + (int64_t)(j * DS_T + tid) * V;
			if constexpr (ALIGNED) {
				const int64_t kc = k < n ? k : 0;
#pragma unroll
				for (int r = 0; r < R; r++) {
					const vec_t t = (DS_NT && NG >= 0) ? __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(row[r] + kc)) : *reinterpret_cast<const vec_t*>(row[r] + kc);
#pragma unroll
					for (int v = 0; v < V; v++) st[j][r][v] = t[v];
				}
			} else {
#pragma unroll
				for (int r = 0; r < R; r++)
#pragma unroll
					for (int v = 0; v < V; v++) st[j][r][v] = row[r][k + v < n ? k + v : 0];
			}
		}
	};
	request(0);
	const int64_t nslots = (int64_t)ngroups * 64;
	const int pos0 = group0 * 64;  // first position of this pass (positions pos0 .. pos0 + DS_G * DS_T - 1)
	// The 16 groups of 64 positions of a pass are sorted by the length of their lists (the design rows are: nrm_design_plan); a wave takes two of
	// them -- group `wave` and, counted from the other end, group 15 - wave: the longest with the shortest (DS_SNAKE; taken in order, wave 0 walked
	// the two longest of each half and every chunk's barrier waited for it)
	auto group_of = [&](int g) { return (DS_SNAKE && (g & 1)) ? (g + 1) * (DS_T / 64) - 1 - wave : g * (DS_T / 64) + wave; };
	int sl_cur[DS_G], sl_nxt[DS_G];  // the design row (within the pass) this thread's position g gathers for in the current / next chunk, -1: none
#pragma unroll
	for (int g = 0; g < DS_G; g++) {
		const int p = pos0 + group_of(g) * 64 + lane;
		sl_cur[g] = p < nslots ? sig[p] - pos0 : -1;
		sl_nxt[g] = -1;
	}
	// sums of this wave over its cell groups: da[g] = covariates 4 g .. 4 g + 3 against the rows on the matrix cores (every lane one entry of four
	// 4 x 4 blocks: cells = b mod 4); dq = |y|^2 and dk = the plain sum (the constant covariate) of the lane's row (lane & 3) over the lane's cells
	double da[NG > 0 ? NG : 1], dq = 0.0, dk = 0.0;
#pragma unroll
	for (int g = 0; g < (NG > 0 ? NG : 1); g++) da[g] = 0.0;
	auto matrix_sums = [&](int c) {
			// The wave's 16 cell groups in batches of DS_UB: the covariate operands of a batch are requested (from L2: every workgroup reads all of
			// d_ct once), the rows' operands read from LDS meanwhile.  BEFORE the next chunk's rows are requested: those loads must stay in flight
			// through the gathers, and the memory counter is in order -- a covariate load behind them could only be waited for together with them.
			// (All 16 operands at once cost 32 registers: 80 + 108 bytes of scratch at the six waves per SIMD this kernel lives on.)
			constexpr int U = DS_CH / 16 / (DS_T / 64);
			// (one base per operand and constant offsets from it: with an address per cell group the compiler kept 16 of each across the chunk loop -- in scratch)
			const double* ctw = ct + ((int64_t)c * DS_CH + wave * U * 16) * 4 + lane;
			const T* lw = reinterpret_cast<const T*>(lds) + (sizeof(T) == 4 ? wave * U * 64 + lane : (wave * U * 16 + (lane >> 2)) * 2 + (lane & 1));
#pragma unroll
			for (int h = 0; h < U; h += DS_UB) {
				__builtin_amdgcn_sched_barrier(0);  // (the batches one after the other: hoisted together their operands spill)
				double cv[NG > 0 ? NG : 1][DS_UB];
				if constexpr (NG > 0) {
#pragma unroll
					for (int g = 0; g < NG; g++)
#pragma unroll
						for (int u = 0; u < DS_UB; u++) cv[g][u] = ctw[(int64_t)g * ctstride + (h + u) * 64];
				}
#pragma unroll
				for (int u = 0; u < DS_UB; u++) {
					double yd;
					if constexpr (sizeof(T) == 4)
						yd = (double)lw[(h + u) * 64];  // row lane & 3 of cell 16 (wave U + h + u) + (lane >> 2)
					else
						yd = (lane & 2) ? 0.0 : (double)lw[(h + u) * 32];  // (two rows per workgroup)
					// |y|^2 and the plain sum (the constant covariate) on the vector ALU: two fp64 operations per lane and cell group.  As matrix
					// instructions (A = the rows themselves, of which only the diagonal is wanted; A = a constant row) they cost 16 cycles each of
					// the SAME fp64 units the gathers' conversions and additions run on -- the fp64 matrix rate of this chip is its vector rate --:
					// k_de_sparse 1.61 -> 2.00 ms with them, against 8 cycles for these two
					dq = fma(yd, yd, dq);
					dk += yd;
					asm volatile("" : "+v"(dq), "+v"(dk));  // here, not later: left to the scheduler the two sums were taken after the section, all 16 converted values kept
					                                        // until then -- and the next chunk's rows, just requested, waited for and spilled to make room
#pragma unroll
					for (int g = 0; g < NG; g++) da[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(cv[g][u], yd, da[g], 0, 0, 0);
				}
			}
			__builtin_amdgcn_sched_barrier(0);
	};
	for (int c = 0; c < nchunks; c++) {
		__syncthreads();  // the gathers of the previous chunk are done with LDS
		if (c > 0) {
			// Position p of the workgroup gathers, in chunk c, for the design row sig[c][p]: the rows are dealt anew for every chunk, sorted by
			// the number of entries they have IN it, so that the 64 lists a wave walks in step are equally long (dealt once for all chunks a
			// wave waited for its longest list: 750 000 padded entries for 500 000, now 570 000).  The sums move with the rows: through
			// LDS, which is free between two chunks.  (sl_cur / sl_nxt were fetched while the previous chunk was gathered: fetched here,
			// between the barriers, their latency was paid twice per chunk by every wave at once and ate what the shorter lists save.)
			double* acc = reinterpret_cast<double*>(lds);  // [design row of this pass][R]
#pragma unroll
			for (int g = 0; g < DS_G; g++)
				if (sl_cur[g] >= 0) {
#pragma unroll
					for (int r = 0; r < R; r++) acc[sl_cur[g] * R + r] = S[g][r];
				}
			__syncthreads();
#pragma unroll
			for (int g = 0; g < DS_G; g++) {
				if (sl_nxt[g] >= 0) {
#pragma unroll
					for (int r = 0; r < R; r++) S[g][r] = acc[sl_nxt[g] * R + r];
				}
				sl_cur[g] = sl_nxt[g];
			}
			__syncthreads();
		}
		const int64_t k0 = (int64_t)c * DS_CH;
#pragma unroll
		for (int j = 0; j < NJ; j++) {
			const int cell = (j * DS_T + tid) * V;
#pragma unroll
			for (int v = 0; v < V; v++) {
				rec_t o;
#pragma unroll
				for (int r = 0; r < R; r++) o[r] = k0 + cell + v < n ? st[j][r][v] : (T)0;
				lds[cell + v] = o;
			}
		}
		__syncthreads();
		if constexpr (NG >= 0 && !DS_MFMA_AFTER) matrix_sums(c);
		if (c + 1 < nchunks) {
			request(c + 1);
#pragma unroll
			for (int g = 0; g < DS_G; g++) {
				const int p = pos0 + group_of(g) * 64 + lane;
				sl_nxt[g] = p < nslots ? sig[(int64_t)(c + 1) * nslots + p] - pos0 : -1;
			}
		}
		// the design rows' cells of this chunk: 8 entries per lane and load (ix_t), the next 8 requested before these are gathered
#pragma unroll
		for (int g = 0; g < DS_G; g++) {
			const int grp = group0 + group_of(g);
			if (grp >= ngroups) continue;
			const int nb = ellw[(int64_t)c * ngroups + grp] >> 3;  // blocks of 8 entries (widths are multiples of 8)
			if (nb == 0) continue;
			const int64_t base = ellbase[(int64_t)c * ngroups + grp] + lane * 8;
			ix_t cur = *reinterpret_cast<const ix_t*>(ell + base);
			for (int jb = 0; jb < nb; jb++) {
				const int64_t at = base + (int64_t)(jb + 1 < nb ? jb + 1 : jb) * 512;
				const ix_t nxt = *reinterpret_cast<const ix_t*>(ell + at);
#pragma unroll
				for (int u = 0; u < 8; u++) {
					const rec_t rec = lds[(int)cur[u]];
					if constexpr (BINARY) {
#pragma unroll
						for (int r = 0; r < R; r++) S[g][r] += (double)rec[r];
					} else {
						const double val = ellv[base + (int64_t)jb * 512 + u];
#pragma unroll
						for (int r = 0; r < R; r++) S[g][r] = fma((double)rec[r], val, S[g][r]);
					}
				}
				cur = nxt;
			}
		}
		if constexpr (NG >= 0 && DS_MFMA_AFTER) matrix_sums(c);
	}
	// the rows' products with the covariates a = y C^T and |y|^2: from the matrix cores above, or as the stream kernel of nrm_single1.hip / an earlier
	// pass left them (common[c * ny + y], row nc: |y|^2)
	__shared__ double as[R][DS_NCMAX];
	__shared__ double yraw[R];
	__syncthreads();
	if constexpr (NG >= 0) {
		// the four blocks of an instruction hold the cells = 0, 1, 2, 3 mod 4: added up across the lanes that differ in bits 2 and 3, then the
		// eight waves in order (through LDS, which the gathers are done with): wsum[wave][which][4 i + j], which = groups, |y|^2, constant
		double* wsum = reinterpret_cast<double*>(lds);
		auto fold = [&](double v, int which) {
			v += __shfl_xor(v, 4, 64);
			v += __shfl_xor(v, 8, 64);
			if ((lane & 12) == 0) wsum[(wave * (NG + 2) + which) * 16 + (lane >> 4) * 4 + (lane & 3)] = v;
		};
#pragma unroll
		for (int g = 0; g < NG; g++) fold(da[g], g);
		// |y|^2 and the plain sums: per lane over its cells; all lanes of a row (lane & 3) together, kept where a product with covariate 0 would sit
		auto fold_rows = [&](double v, int which) {
#pragma unroll
			for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
			if (lane < 4) wsum[(wave * (NG + 2) + which) * 16 + lane] = v;
		};
		fold_rows(dq, NG);
		fold_rows(dk, NG + 1);
		__syncthreads();
		if (tid < R) {
			const int r = tid;
			auto total = [&](int which, int i) {
				double t = 0.0;
				for (int w = 0; w < DS_T / 64; w++) t += wsum[(w * (NG + 2) + which) * 16 + i * 4 + r];
				return t;
			};
			int m = 0;  // covariates other than the constant one, in their order: 4 per group
			for (int c = 0; c < nc; c++) {
				if (c == cidx)
					as[r][c] = cval * total(NG + 1, 0);
				else {
					as[r][c] = total(m >> 2, m & 3);
					m++;
				}
			}
			yraw[r] = total(NG, 0);
			if (y0 + r < ny) {  // for the kernel's later passes (more than DS_PASS design rows)
				for (int c = 0; c < nc; c++) common[(int64_t)c * ny + y0 + r] = as[r][c];
				common[(int64_t)nc * ny + y0 + r] = yraw[r];
			}
		}
	} else if (tid < R && y0 + tid < ny) {
		for (int c = 0; c < nc; c++) as[tid][c] = common[(int64_t)c * ny + y0 + tid];
		yraw[tid] = common[(int64_t)nc * ny + y0 + tid];
	}
	if (tid < R && y0 + tid < ny) {
		const int r = tid;
		const int64_t y = y0 + r;
		double yy = yraw[r];
		if (first) {
			// b_y = a (C C^T)^+ (association.py:227-229), |y~|^2 = |y|^2 - a . b_y
			for (int c = 0; c < nc; c++) {
				double b = 0.0;
				for (int e = 0; e < nc; e++) b = fma(as[r][e], dci[e * nc + c], b);
				if (coefy) coefy[y * nc + c] = b;
				yy = fma(-as[r][c], b, yy);
			}
			ssy[y] = yy > 0.0 ? yy : 0.0;
			// |y~|^2 as a difference: a row whose residual is less than a hundredth of the row itself (|y~|^2 < 1e-4 |y|^2) has lost four of
			// fp64's sixteen digits here, and its products with the design rows likewise -- counted like a pair the integer engine
			// cannot certify (flags[2]): the caller redoes such a call on K1's two sweeps and the fp64 Gram kernel
			if (flags && !(yy >= 1e-4 * yraw[r]) && yraw[r] > 0.0) atomicAdd(&flags[2], 1);
		}
	}
	__syncthreads();
	// y~ . x~ for this thread's design rows
#pragma unroll
	for (int g = 0; g < DS_G; g++) {
		if (group0 + group_of(g) >= ngroups) continue;
		const int x = sl_cur[g] >= 0 ? slot2x[pos0 + sl_cur[g]] : -1;  // (the design row this position gathered for in the last chunk)
		if (x < 0) continue;
#pragma unroll
		for (int r = 0; r < R; r++) {
			double d = S[g][r];
			for (int c = 0; c < nc; c++) d = fma(-as[r][c], bx[(int64_t)x * ldb + c], d);
			if (y0 + r < ny) dot[by_gene ? (y0 + r) * ldd + x : (int64_t)x * ldd + y0 + r] = d;
		}
	}
}

// the covariates cell by cell in fours, as the A operand of v_mfma_f64_4x4x4_4b_f64 reads them: ct[g][cell][i] = C[map(4 g + i)][cell], map = the
// covariates in their order without the constant one (cidx); zero for cells >= n (up to whole chunks) and for the last group's empty places
__global__ void __launch_bounds__(256) k_ds_ct(const double* __restrict__ C, int64_t ldc, int nc, int cidx, int64_t n, int64_t ncells, int ng, double* __restrict__ ct) {
	const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (group, cell, i)
	if (e >= (int64_t)ng * ncells * 4) return;
	const int i = (int)(e & 3);
	const int64_t cell = (e >> 2) % ncells;
	const int g = (int)((e >> 2) / ncells);
	int m = g * 4 + i;  // the m-th covariate that is not the constant one
	if (cidx >= 0 && m >= cidx) m++;
	ct[e] = (m < nc && cell < n) ? C[(int64_t)m * ldc + cell] : 0.0;
}

template <typename T, bool BINARY>
int ds_go(const void* d_y, int64_t ldy, int64_t n, int64_t ny, double* d_common, int nc, const double* d_dci, const int16_t* d_ell, const double* d_ellv,
		  const int64_t* d_base, const int32_t* d_w, const int32_t* d_sig, int ngroups, const int32_t* d_slot2x, const double* d_bx, int64_t ldb, double* d_dot, int64_t ldd,
		  int by_gene, double* d_ssy, double* d_coefy, int32_t* d_flags, const double* d_ct, int64_t ctstride, int ng, int cidx, double cval, hipStream_t st) {
	constexpr int R = DsVec<T>::R;
	const bool aligned = ((uintptr_t)d_y % 16 == 0) && (ldy * sizeof(T)) % 16 == 0 && n % DsVec<T>::V == 0;
	const dim3 grid((unsigned)((ny + R - 1) / R));
#define DS_LAUNCH(AL, NGV)                                                                                                                                         \
	hipLaunchKernelGGL((k_de_sparse<T, AL, BINARY, NGV>), grid, dim3(DS_T), 0, st, (const T*)d_y, ldy, n, ny, d_common, nc, d_dci, d_ell, d_ellv, d_base, d_w, d_sig, \
					   ngroups, g0, d_slot2x, d_bx, ldb, d_dot, ldd, by_gene, d_ssy, d_coefy, d_flags, g0 == 0 ? 1 : 0, d_ct, ctstride, cidx, cval)
	for (int g0 = 0; g0 < ngroups; g0 += (DS_T / 64) * DS_G) {  // 1024 design rows per pass; the first takes the sums with the covariates (ng >= 0)
		const int ngv = g0 == 0 ? ng : -1;
		if (aligned) {
			switch (ngv) {
				case 0: DS_LAUNCH(true, 0); break;
				case 1: DS_LAUNCH(true, 1); break;
				case 2: DS_LAUNCH(true, 2); break;
				default: DS_LAUNCH(true, -1); break;
			}
		} else {
			switch (ngv) {
				case 0: DS_LAUNCH(false, 0); break;
				case 1: DS_LAUNCH(false, 1); break;
				case 2: DS_LAUNCH(false, 2); break;
				default: DS_LAUNCH(false, -1); break;
			}
		}
	}
#undef DS_LAUNCH
	return nrm_check_launch("k_de_sparse");
}

}  // namespace

extern "C" int64_t nrm_de_sparse_chunk(void) { return DS_CH; }
extern "C" int64_t nrm_de_sparse_max_covariates(void) { return DS_NCMAX; }
// covariates (besides a constant one) up to which nrm_de_sparse takes the rows' products with them itself (d_ct given)
extern "C" int64_t nrm_de_sparse_fused_covariates(void) { return DS_FUSED_NC; }
// doubles of the d_ct scratch of nrm_de_sparse for n cells and nc covariates of which one is constant (const_idx >= 0) or none
extern "C" int64_t nrm_de_sparse_ct_doubles(int64_t n, int64_t nc, int64_t const_idx) {
	const int64_t m = nc - (const_idx >= 0 ? 1 : 0), ng = (m + 3) / 4, ncells = (n + DS_CH - 1) / DS_CH * DS_CH;
	return ng * ncells * 4 > 0 ? ng * ncells * 4 : 1;
}

extern "C" int nrm_de_sparse(const void* d_y, int y_dtype, int64_t ny, int64_t n, int64_t ldy, double* d_common, int64_t nc, const double* d_dci,
							 const int16_t* d_ell, const double* d_ellv, const int64_t* d_base, const int32_t* d_w, const int32_t* d_sig, int64_t ngroups,
							 const int32_t* d_slot2x, const double* d_bx, int64_t ldb, double* d_dot, int64_t ldd, int by_gene, double* d_ssy, double* d_coefy, int32_t* d_flags,
							 const double* d_c, int64_t ldc, int64_t const_idx, double const_val, double* d_ct, void* stream) {
	NRM_REQUIRE(ny > 0 && n > 0 && nc >= 0 && nc <= DS_NCMAX && ngroups > 0 && ngroups < (1 << 24), "nrm_de_sparse: bad sizes (at most %d covariates)", DS_NCMAX);
	NRM_REQUIRE(y_dtype == NRM_F32 || y_dtype == NRM_F64, "nrm_de_sparse: bad dtype");
	NRM_REQUIRE(ldy >= n && (nc == 0 || ldb >= nc) && (by_gene || ldd >= ny), "nrm_de_sparse: pitch too small");
	NRM_REQUIRE(d_y && d_common && d_ell && d_base && d_w && d_sig && d_slot2x && d_dot && d_ssy && (nc == 0 || (d_dci && d_bx)), "nrm_de_sparse: null pointer");
	hipStream_t st = (hipStream_t)stream;
	// d_ct given: the rows' products with the covariates and their sums of squares are taken inside the kernel (and left in d_common)
	int ng = -1, cidx = -1;
	int64_t ctstride = 0;
	if (d_ct) {
		cidx = (const_idx >= 0 && const_idx < nc) ? (int)const_idx : -1;
		const int64_t m = nc - (cidx >= 0 ? 1 : 0);
		NRM_REQUIRE(m <= DS_FUSED_NC && (nc == 0 || (d_c && ldc >= n)), "nrm_de_sparse: at most %d covariates besides a constant one with d_ct (run nrm_single1_stream first otherwise)", DS_FUSED_NC);
		ng = (int)((m + 3) / 4);
		const int64_t ncells = (n + DS_CH - 1) / DS_CH * DS_CH;
		ctstride = ncells * 4;
		if (ng > 0)
			hipLaunchKernelGGL(k_ds_ct, dim3((unsigned)((ng * ncells * 4 + 255) / 256)), dim3(256), 0, st, d_c, ldc, (int)nc, cidx, n, ncells, ng, d_ct);
	}
#define DS_GO(T, BIN) ds_go<T, BIN>(d_y, ldy, n, ny, d_common, (int)nc, d_dci, d_ell, d_ellv, d_base, d_w, d_sig, (int)ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, by_gene, d_ssy, d_coefy, \
									 d_flags, d_ct, ctstride, ng, cidx, const_val, st)
	if (y_dtype == NRM_F64) return d_ellv ? DS_GO(double, false) : DS_GO(double, true);
	return d_ellv ? DS_GO(float, false) : DS_GO(float, true);
#undef DS_GO
}

// ---- the design rows' own statistics from their entries ---------------------------------------------------------------------------------
// |x~_i|^2 = |x_i|^2 - (x_i C^T) . b_i and b_i = (x_i C^T)(C C^T)^+ (association.py:224-230) need of a sparse design row only its entries:
// a wave per design row walks them (lane-strided, then a tree over the lanes: a fixed order), instead of K1's two sweeps over n cells.
__global__ void __launch_bounds__(64) k_design_stats(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ cells, const double* __restrict__ vals,
													  const double* __restrict__ C, int64_t ldc, int nc, const double* __restrict__ dci, int64_t nx,
													  double* __restrict__ ss, double* __restrict__ coef, int32_t* __restrict__ flags) {
	const int64_t i = blockIdx.x;
	const int lane = threadIdx.x;
	double a[DS_NCMAX], xx = 0.0;
#pragma unroll
	for (int c = 0; c < DS_NCMAX; c++) a[c] = 0.0;
	for (int64_t e = row_ptr[i] + lane; e < row_ptr[i + 1]; e += 64) {
		const double v = vals ? vals[e] : 1.0;
		const int64_t k = cells[e];
		xx = fma(v, v, xx);
#pragma unroll
		for (int c = 0; c < DS_NCMAX; c++)
			if (c < nc) a[c] = fma(v, C[c * ldc + k], a[c]);
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		xx += __shfl_down(xx, o, 64);
#pragma unroll
		for (int c = 0; c < DS_NCMAX; c++)
			if (c < nc) a[c] += __shfl_down(a[c], o, 64);
	}
	if (lane == 0) {
		double s = xx;
		for (int c = 0; c < nc; c++) {
			double b = 0.0;
#pragma unroll
			for (int e = 0; e < DS_NCMAX; e++)
				if (e < nc) b = fma(a[e], dci[e * nc + c], b);
			coef[i * nc + c] = b;
			double ac = 0.0;
#pragma unroll
			for (int e = 0; e < DS_NCMAX; e++)
				if (e == c) ac = a[e];
			s = fma(-ac, b, s);
		}
		ss[i] = s > 0.0 ? s : 0.0;
		// the same difference as on the expression side (k_de_sparse): a design row all but inside the span of the covariates (a gRNA that
		// coincides with a batch indicator) has lost the digits of |x~|^2 -- and its products with the expression rows likewise -- counted in
		// flags[2], and the caller redoes the call on K1's two sweeps and the fp64 Gram kernel, which residualise first
		if (flags && !(s >= 1e-4 * xx) && xx > 0.0) atomicAdd(&flags[2], 1);
	}
}

// d_row_ptr (nx + 1), d_cells (int32), d_vals (fp64 or NULL: every entry 1): the entries of design row i are [d_row_ptr[i], d_row_ptr[i + 1]).
// d_ss (nx) = |x~_i|^2, d_coef (nx, nc) = b_i.  nc = 0: no covariates (d_ss = |x_i|^2).  d_flags (int32[4]) or NULL: [2] counts the rows with |x~|^2 < 1e-4 |x|^2.
extern "C" int nrm_design_stats(const int64_t* d_row_ptr, const int32_t* d_cells, const double* d_vals, const double* d_c, int64_t ldc, int64_t nc,
								const double* d_dci, int64_t nx, double* d_ss, double* d_coef, int32_t* d_flags, void* stream) {
	NRM_REQUIRE(nx > 0 && nc >= 0 && nc <= DS_NCMAX && d_row_ptr && d_cells && d_ss && (nc == 0 || (d_c && d_dci && d_coef)), "nrm_design_stats: bad arguments");
	hipLaunchKernelGGL(k_design_stats, dim3((unsigned)nx), dim3(64), 0, (hipStream_t)stream, d_row_ptr, d_cells, d_vals, d_c, ldc, (int)nc, d_dci, nx, d_ss, d_coef, d_flags);
	return nrm_check_launch("k_design_stats");
}
