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
// pairs.  The entries come in ELL form per (chunk, 64 slots of a wave): entry j of the 64 lanes side by side (one coalesced 128-byte
// load), lists padded to the longest of the 64 with the offset of a record of zeros.  The next chunk's HBM loads are in flight (in
// registers) while the current one is gathered.  Covariate products and sums of squares are taken from the registers on the way into LDS.
#include "nrm_common.h"

#define DS_CH 4096   // cells per chunk: 64 KB of records
#define DS_T 512     // threads per workgroup (one per CU: 64 KB of LDS, up to 256 registers)
#define DS_G 2       // design rows per thread: 1024 per pass over the expression matrix
#define DS_NCMAX 8

namespace {

template <typename T>
struct DsVec;
template <>
struct DsVec<float> {
	static constexpr int V = 4, R = 4;
};
template <>
struct DsVec<double> {
	static constexpr int V = 2, R = 2;
};

template <typename T, int NC, bool ALIGNED, bool BINARY, typename VT>
__global__ void __launch_bounds__(DS_T) k_de_sparse(const T* __restrict__ Y, int64_t ldy, int64_t n, int64_t ny, const double* __restrict__ C, int64_t ldc,
													   const double* __restrict__ dci, const int16_t* __restrict__ ell, const VT* __restrict__ ellv,
													   const int64_t* __restrict__ ellbase, const int32_t* __restrict__ ellw, int ngroups, int group0,
													   const int32_t* __restrict__ slot2x, const double* __restrict__ bx, int64_t ldb,
													   double* __restrict__ dot, int64_t ldd, double* __restrict__ ssy, double* __restrict__ coefy, int first) {
	constexpr int V = DsVec<T>::V, R = DsVec<T>::R;
	constexpr int NJ = DS_CH / (DS_T * V);  // groups of V consecutive cells a thread stages per chunk
	constexpr int NCA = NC > 0 ? NC : 1;
	typedef T rec_t __attribute__((ext_vector_type(R)));
	typedef T vec_t __attribute__((ext_vector_type(V)));
	__shared__ rec_t lds[DS_CH + 1];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int64_t y0 = (int64_t)blockIdx.x * R;
	const T* row[R];
#pragma unroll
	for (int r = 0; r < R; r++) row[r] = Y + (y0 + r < ny ? y0 + r : ny - 1) * ldy;
	double S[DS_G][R], a[R][NCA], q[R];
#pragma unroll
	for (int r = 0; r < R; r++) {
		q[r] = 0.0;
#pragma unroll
		for (int c = 0; c < NC; c++) a[r][c] = 0.0;
#pragma unroll
		for (int g = 0; g < DS_G; g++) S[g][r] = 0.0;
	}
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
	for (int c = 0; c < nchunks; c++) {
		__syncthreads();  // the gathers of the previous chunk are done with LDS
		const int64_t k0 = (int64_t)c * DS_CH;
#pragma unroll
		for (int j = 0; j < NJ; j++) {
			const int cell = (j * DS_T + tid) * V;
			const int64_t k = k0 + cell;
#pragma unroll
			for (int v = 0; v < V; v++)
				if (k + v >= n) {
#pragma unroll
					for (int r = 0; r < R; r++) st[j][r][v] = (T)0;
				}
			if (NC > 0) {
				double cv[NCA][V];
				if constexpr (ALIGNED) {
					const int64_t kc = k < n ? k : 0;
#pragma unroll
					for (int cc = 0; cc < NC; cc++)
#pragma unroll
						for (int h = 0; h < V; h += 2) {
							const double2 t = *reinterpret_cast<const double2*>(C + cc * ldc + kc + h);
							cv[cc][h] = t.x;
							cv[cc][h + 1] = t.y;
						}
				} else {
#pragma unroll
					for (int cc = 0; cc < NC; cc++)
#pragma unroll
						for (int h = 0; h < V; h++) cv[cc][h] = C[cc * ldc + (k + h < n ? k + h : 0)];
				}
#pragma unroll
				for (int v = 0; v < V; v++)
#pragma unroll
					for (int r = 0; r < R; r++) {
						const double yd = (double)st[j][r][v];
#pragma unroll
						for (int cc = 0; cc < NC; cc++) a[r][cc] = fma(yd, cv[cc][v], a[r][cc]);
					}
			}
#pragma unroll
			for (int v = 0; v < V; v++) {
				rec_t o;
#pragma unroll
				for (int r = 0; r < R; r++) {
					const double yd = (double)st[j][r][v];
					q[r] = fma(yd, yd, q[r]);
					o[r] = st[j][r][v];
				}
				lds[cell + v] = o;
			}
		}
		__syncthreads();
		if (c + 1 < nchunks) request(c + 1);
		// the design rows' cells of this chunk
#pragma unroll
		for (int g = 0; g < DS_G; g++) {
			const int grp = group0 + g * (DS_T / 64) + wave;
			if (grp >= ngroups) break;
			const int64_t base = ellbase[(int64_t)c * ngroups + grp] + lane;
			const int w = ellw[(int64_t)c * ngroups + grp];
			int j = 0;
			for (; j + 4 <= w; j += 4) {
				int off[4];
				VT val[4];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					off[u] = ell[base + (int64_t)(j + u) * 64];
					if (!BINARY) val[u] = ellv[base + (int64_t)(j + u) * 64];
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const rec_t rec = lds[off[u]];
#pragma unroll
					for (int r = 0; r < R; r++) S[g][r] = BINARY ? S[g][r] + (double)rec[r] : fma((double)rec[r], (double)val[u], S[g][r]);
				}
			}
			for (; j < w; j++) {
				const int off = ell[base + (int64_t)j * 64];
				const rec_t rec = lds[off];
#pragma unroll
				for (int r = 0; r < R; r++) S[g][r] = BINARY ? S[g][r] + (double)rec[r] : fma((double)rec[r], (double)ellv[base + (int64_t)j * 64], S[g][r]);
			}
		}
	}
	// sums of the workgroup's threads (products with the covariates, sums of squares): within a wave, then the waves in order
	__shared__ double red[DS_T / 64][R * (NCA + 1)], as[R][NCA];
#pragma unroll
	for (int r = 0; r < R; r++) {
#pragma unroll
		for (int c = 0; c < NC; c++) {
			double t = a[r][c];
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
			if (lane == 0) red[wave][r * (NCA + 1) + c] = t;
		}
		double t = q[r];
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
		if (lane == 0) red[wave][r * (NCA + 1) + NCA] = t;
	}
	__syncthreads();
	if (tid < R) {
		const int r = tid;
		double av[NCA];
#pragma unroll
		for (int c = 0; c < NC; c++) {
			double t = 0.0;
#pragma unroll
			for (int w = 0; w < DS_T / 64; w++) t += red[w][r * (NCA + 1) + c];
			av[c] = t;
			as[r][c] = t;
		}
		double yy = 0.0;
#pragma unroll
		for (int w = 0; w < DS_T / 64; w++) yy += red[w][r * (NCA + 1) + NCA];
		if (first && y0 + r < ny) {
			// b_y = a (C C^T)^+ (association.py:227-229), |y~|^2 = |y|^2 - a . b_y
#pragma unroll
			for (int c = 0; c < NC; c++) {
				double b = 0.0;
#pragma unroll
				for (int e = 0; e < NC; e++) b = fma(av[e], dci[e * NC + c], b);
				if (coefy) coefy[(y0 + r) * NC + c] = b;
				yy = fma(-av[c], b, yy);
			}
			ssy[y0 + r] = yy > 0.0 ? yy : 0.0;
		}
	}
	__syncthreads();
	// y~ . x~ for this thread's design rows
#pragma unroll
	for (int g = 0; g < DS_G; g++) {
		const int slot = (group0 + g * (DS_T / 64)) * 64 + tid;
		if ((group0 + g * (DS_T / 64) + wave) >= ngroups) break;
		const int x = slot2x[slot];
		if (x < 0) continue;
		double bxv[NCA];
#pragma unroll
		for (int c = 0; c < NC; c++) bxv[c] = bx[(int64_t)x * ldb + c];
#pragma unroll
		for (int r = 0; r < R; r++) {
			double d = S[g][r];
#pragma unroll
			for (int c = 0; c < NC; c++) d = fma(-as[r][c], bxv[c], d);
			if (y0 + r < ny) dot[(int64_t)x * ldd + y0 + r] = d;
		}
	}
}

template <typename T, int NC, bool BINARY, typename VT>
void ds_go(const void* d_y, int64_t ldy, int64_t n, int64_t ny, const double* d_c, int64_t ldc, const double* d_dci, const int16_t* d_ell, const void* d_ellv,
		   const int64_t* d_base, const int32_t* d_w, int ngroups, const int32_t* d_slot2x, const double* d_bx, int64_t ldb, double* d_dot, int64_t ldd,
		   double* d_ssy, double* d_coefy, hipStream_t st) {
	constexpr int R = DsVec<T>::R;
	const bool aligned = ((uintptr_t)d_y % 16 == 0) && (ldy * sizeof(T)) % 16 == 0 && n % DsVec<T>::V == 0 && (NC == 0 || (((uintptr_t)d_c % 16 == 0) && ldc % 2 == 0));
	const dim3 grid((unsigned)((ny + R - 1) / R));
	for (int g0 = 0; g0 < ngroups; g0 += (DS_T / 64) * DS_G) {  // 1024 design rows per pass
		if (aligned)
			hipLaunchKernelGGL((k_de_sparse<T, NC, true, BINARY, VT>), grid, dim3(DS_T), 0, st, (const T*)d_y, ldy, n, ny, d_c, ldc, d_dci, d_ell, (const VT*)d_ellv, d_base,
							   d_w, ngroups, g0, d_slot2x, d_bx, ldb, d_dot, ldd, d_ssy, d_coefy, g0 == 0 ? 1 : 0);
		else
			hipLaunchKernelGGL((k_de_sparse<T, NC, false, BINARY, VT>), grid, dim3(DS_T), 0, st, (const T*)d_y, ldy, n, ny, d_c, ldc, d_dci, d_ell, (const VT*)d_ellv, d_base,
							   d_w, ngroups, g0, d_slot2x, d_bx, ldb, d_dot, ldd, d_ssy, d_coefy, g0 == 0 ? 1 : 0);
	}
}

template <typename T, int NC>
void ds_values(int v_dtype, const void* d_y, int64_t ldy, int64_t n, int64_t ny, const double* d_c, int64_t ldc, const double* d_dci, const int16_t* d_ell,
			   const void* d_ellv, const int64_t* d_base, const int32_t* d_w, int ngroups, const int32_t* d_slot2x, const double* d_bx, int64_t ldb, double* d_dot,
			   int64_t ldd, double* d_ssy, double* d_coefy, hipStream_t st) {
	if (v_dtype < 0)
		ds_go<T, NC, true, double>(d_y, ldy, n, ny, d_c, ldc, d_dci, d_ell, nullptr, d_base, d_w, ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, d_ssy, d_coefy, st);
	else
		ds_go<T, NC, false, double>(d_y, ldy, n, ny, d_c, ldc, d_dci, d_ell, d_ellv, d_base, d_w, ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, d_ssy, d_coefy, st);
}

template <typename T>
int ds_nc(int64_t nc, int v_dtype, const void* d_y, int64_t ldy, int64_t n, int64_t ny, const double* d_c, int64_t ldc, const double* d_dci, const int16_t* d_ell,
		  const void* d_ellv, const int64_t* d_base, const int32_t* d_w, int ngroups, const int32_t* d_slot2x, const double* d_bx, int64_t ldb, double* d_dot,
		  int64_t ldd, double* d_ssy, double* d_coefy, hipStream_t st) {
#define DS_CASE(NC)                                                                                                                                       \
	case NC:                                                                                                                                              \
		ds_values<T, NC>(v_dtype, d_y, ldy, n, ny, d_c, ldc, d_dci, d_ell, d_ellv, d_base, d_w, ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, d_ssy, d_coefy, st); \
		break;
	switch ((int)nc) {
		DS_CASE(0)
		DS_CASE(1)
		DS_CASE(2)
		DS_CASE(3)
		DS_CASE(4)
		DS_CASE(5)
		DS_CASE(6)
		DS_CASE(7)
		DS_CASE(8)
	}
#undef DS_CASE
	return nrm_check_launch("k_de_sparse");
}

}  // namespace

extern "C" int64_t nrm_de_sparse_chunk(void) { return DS_CH; }
extern "C" int64_t nrm_de_sparse_max_covariates(void) { return DS_NCMAX; }

extern "C" int nrm_de_sparse(const void* d_y, int y_dtype, int64_t ny, int64_t n, int64_t ldy, const double* d_c, int64_t nc, int64_t ldc, const double* d_dci,
							 const int16_t* d_ell, const void* d_ellv, int v_dtype, const int64_t* d_base, const int32_t* d_w, int64_t ngroups,
							 const int32_t* d_slot2x, const double* d_bx, int64_t ldb, double* d_dot, int64_t ldd, double* d_ssy, double* d_coefy, void* stream) {
	NRM_REQUIRE(ny > 0 && n > 0 && nc >= 0 && nc <= DS_NCMAX && ngroups > 0 && ngroups < (1 << 24), "nrm_de_sparse: bad sizes (at most %d covariates)", DS_NCMAX);
	NRM_REQUIRE(y_dtype == NRM_F32 || y_dtype == NRM_F64, "nrm_de_sparse: bad dtype");
	NRM_REQUIRE(v_dtype < 0 || v_dtype == NRM_F64, "nrm_de_sparse: values are fp64 (or absent: every entry 1)");
	NRM_REQUIRE(ldy >= n && (nc == 0 || (ldc >= n && ldb >= nc)) && ldd >= ny, "nrm_de_sparse: pitch too small");
	NRM_REQUIRE(d_y && d_ell && d_base && d_w && d_slot2x && d_dot && d_ssy && (nc == 0 || (d_c && d_dci && d_bx)) && (v_dtype < 0 || d_ellv),
				"nrm_de_sparse: null pointer");
	hipStream_t st = (hipStream_t)stream;
	if (y_dtype == NRM_F64)
		return ds_nc<double>(nc, v_dtype, d_y, ldy, n, ny, d_c, ldc, d_dci, d_ell, d_ellv, d_base, d_w, (int)ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, d_ssy, d_coefy, st);
	return ds_nc<float>(nc, v_dtype, d_y, ldy, n, ny, d_c, ldc, d_dci, d_ell, d_ellv, d_base, d_w, (int)ngroups, d_slot2x, d_bx, ldb, d_dot, ldd, d_ssy, d_coefy, st);
}
