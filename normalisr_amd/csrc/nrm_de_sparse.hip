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
// The next chunk's HBM loads are in flight (in registers) while the current one is gathered.  The products with the covariates y C^T and
// |y|^2 come from nrm_single1.hip's stream kernel (sums only: a second pass over the rows at HBM rate, 0.6 ms at configs[3] size; taken
// inside THIS kernel they cost more both ways it was tried: covariate values fetched between the barriers of a chunk, every wave in the same
// phase: +1.1 ms; a dense phase after the gathers -- consecutive lanes on consecutive records and covariate values, two cells at a time to stay
// inside the registers: +2.0 ms, each batch waiting for its loads at two waves per SIMD; the covariate values travelling with the rows, requested a
// chunk ahead: +1.0 ms -- a workgroup of 4 rows re-reads the covariates once per 4 rows, 7.5 GB through L2 beside the 3 GB of rows, where the
// stream kernel's 8 rows per workgroup and its own pass cost 0.64).
#include "nrm_common.h"
#include "nrm_design.h"  // DS_CH (cells per chunk), DS_T (threads per workgroup), DS_G (design rows per thread)

#define DS_NCMAX 32  // (the sums over the covariates come from nrm_single1_stream: its limit)
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

template <typename T, bool ALIGNED, bool BINARY>
__global__ void __launch_bounds__(DS_T, DS_WGS) k_de_sparse(const T* __restrict__ Y, int64_t ldy, int64_t n, int64_t ny, const double* __restrict__ common, int nc,
															   const double* __restrict__ dci, const int16_t* __restrict__ ell, const double* __restrict__ ellv,
															   const int64_t* __restrict__ ellbase, const int32_t* __restrict__ ellw, const int32_t* __restrict__ sig, int ngroups, int group0,
															   const int32_t* __restrict__ slot2x, const double* __restrict__ bx, int64_t ldb,
															   double* __restrict__ dot, int64_t ldd, int by_gene, double* __restrict__ ssy, double* __restrict__ coefy, int32_t* __restrict__ flags, int first) {
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
					const vec_t t = *reinterpret_cast<const vec_t*>(row[r] + kc);
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
	int sl_cur[DS_G], sl_nxt[DS_G];  // the design row (within the pass) this thread's position g gathers for in the current / next chunk, -1: none
#pragma unroll
	for (int g = 0; g < DS_G; g++) {
		const int p = pos0 + g * DS_T + tid;
		sl_cur[g] = p < nslots ? sig[p] - pos0 : -1;
		sl_nxt[g] = -1;
	}
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
		if (c + 1 < nchunks) {
			request(c + 1);
#pragma unroll
			for (int g = 0; g < DS_G; g++) {
				const int p = pos0 + g * DS_T + tid;
				sl_nxt[g] = p < nslots ? sig[(int64_t)(c + 1) * nslots + p] - pos0 : -1;
			}
		}
		// the design rows' cells of this chunk: 8 entries per lane and load (ix_t), the next 8 requested before these are gathered
#pragma unroll
		for (int g = 0; g < DS_G; g++) {
			const int grp = group0 + g * (DS_T / 64) + wave;
			if (grp >= ngroups) break;
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
	}
	// the rows' products with the covariates a = y C^T and |y|^2 were summed by the stream kernel of nrm_single1.hip (common[c * ny + y], row nc: |y|^2)
	__shared__ double as[R][DS_NCMAX];
	__syncthreads();
	if (tid < R && y0 + tid < ny) {
		const int r = tid;
		const int64_t y = y0 + r;
		double yy = common[(int64_t)nc * ny + y];
		for (int c = 0; c < nc; c++) as[r][c] = common[(int64_t)c * ny + y];
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
			if (flags && !(yy >= 1e-4 * common[(int64_t)nc * ny + y]) && common[(int64_t)nc * ny + y] > 0.0) atomicAdd(&flags[2], 1);
		}
	}
	__syncthreads();
	// y~ . x~ for this thread's design rows
#pragma unroll
	for (int g = 0; g < DS_G; g++) {
		const int p = (group0 + g * (DS_T / 64)) * 64 + tid;
		if ((group0 + g * (DS_T / 64) + wave) >= ngroups) break;
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

template <typename T, bool BINARY>
int ds_go(const void* d_y, int64_t ldy, int64_t n, int64_t ny, const double* d_common, int nc, const double* d_dci, const int16_t* d_ell, const double* d_ellv,
		  const int64_t* d_base, const int32_t* d_w, const int32_t* d_sig, int ngroups, const int32_t* d_slot2x, const double* d_bx, int64_t ldb, double* d_dot, int64_t ldd,
		  int by_gene, double* d_ssy, double* d_coefy, int32_t* d_flags, hipStream_t st) {
	constexpr int R = DsVec<T>::R;
	const bool aligned = ((uintptr_t)d_y % 16 == 0) && (ldy * sizeof(T)) % 16 == 0 && n % DsVec<T>::V == 0;
	const dim3 grid((unsigned)((ny + R - 1) / R));
	for (int g0 = 0; g0 < ngroups; g0 += (DS_T / 64) * DS_G) {  // 1024 design rows per pass
		if (aligned)
			hipLaunchKernelGGL((k_de_sparse<T, true, BINARY>), grid, dim3(DS_T), 0, st, (const T*)d_y, ldy, n, ny, d_common, nc, d_dci, d_ell, d_ellv, d_base, d_w,
							   d_sig, ngroups, g0, d_slot2x, d_bx, ldb, d_dot, ldd, by_gene, d_ssy, d_coefy, d_flags, g0 == 0 ? 1 : 0);
		else
			hipLaunchKernelGGL((k_de_sparse<T, false, BINARY>), grid, dim3(DS_T), 0, st, (const T*)d_y, ldy, n, ny, d_common, nc, d_dci, d_ell, d_ellv, d_base, d_w,
							   d_sig, ngroups, g0, d_slot2x, d_bx, ldb, d_dot, ldd, by_gene, d_ssy, d_coefy, d_flags, g0 == 0 ? 1 : 0);
	}
	return nrm_check_launch("k_de_sparse");
}

}  // namespace

extern "C" int64_t nrm_de_sparse_chunk(void) { return DS_CH; }
extern "C" int64_t nrm_de_sparse_max_covariates(void) { return DS_NCMAX; }

extern "C" int nrm_de_sparse(const void* d_y, int y_dtype, int64_t ny, int64_t n, int64_t ldy, const double* d_common, int64_t nc, const double* d_dci,
							 const int16_t* d_ell, const double* d_ellv, const int64_t* d_base, const int32_t* d_w, const int32_t* d_sig, int64_t ngroups,
							 const int32_t* d_slot2x, const double* d_bx, int64_t ldb, double* d_dot, int64_t ldd, int by_gene, double* d_ssy, double* d_coefy, int32_t* d_flags, void* stream) {
	NRM_REQUIRE(ny > 0 && n > 0 && nc >= 0 && nc <= DS_NCMAX && ngroups > 0 && ngroups < (1 << 24), "nrm_de_sparse: bad sizes (at most %d covariates)", DS_NCMAX);
	NRM_REQUIRE(y_dtype == NRM_F32 || y_dtype == NRM_F64, "nrm_de_sparse: bad dtype");
	NRM_REQUIRE(ldy >= n && (nc == 0 || ldb >= nc) && (by_gene || ldd >= ny), "nrm_de_sparse: pitch too small");
	NRM_REQUIRE(d_y && d_common && d_ell && d_base && d_w && d_sig && d_slot2x && d_dot && d_ssy && (nc == 0 || (d_dci && d_bx)), "nrm_de_sparse: null pointer");
	hipStream_t st = (hipStream_t)stream;
	if (y_dtype == NRM_F64)
		return d_ellv ? ds_go<double, false>(d_y, ldy, n, ny, d_common, (int)nc, d_dci, d_ell, d_ellv, d_base, d_w, d_sig, (int)ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, by_gene, d_ssy, d_coefy, d_flags, st)
					  : ds_go<double, true>(d_y, ldy, n, ny, d_common, (int)nc, d_dci, d_ell, d_ellv, d_base, d_w, d_sig, (int)ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, by_gene, d_ssy, d_coefy, d_flags, st);
	return d_ellv ? ds_go<float, false>(d_y, ldy, n, ny, d_common, (int)nc, d_dci, d_ell, d_ellv, d_base, d_w, d_sig, (int)ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, by_gene, d_ssy, d_coefy, d_flags, st)
				  : ds_go<float, true>(d_y, ldy, n, ny, d_common, (int)nc, d_dci, d_ell, d_ellv, d_base, d_w, d_sig, (int)ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, by_gene, d_ssy, d_coefy, d_flags, st);
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
