// single=1 ("only cells without any other grouping", low-MOI CRISPR screens; association.py:263-390,911-925).
// Every grouping x_i is tested on its own subset of cells S_i.  All per-(i, gene) quantities are bilinear in the
// gene's expression row, so ONE Gram contraction of Y with the masked rows W_i = [1_S C; 1_S x_i] (K2) and one
// of Y^2 with the masks give the sufficient statistics; this sweep finishes each pair:
//     a = y C_S^T (nc), xy = y.x_S, q = |y_S|^2           (from the Gram matrices)
//     ccy = M_i^+ a                                        association.py:357 (M_i = C_S C_S^T, pseudo-inverse on the host)
//     |y~|^2 = q - a.ccy,  x~.y~ = xy - a.ccx_i            association.py:358-360 in closed form
//     gamma = x~.y~ / (ns vx),  R^2 = gamma^2 vx / vy      association.py:367-371,  dof_i = ns_i - 1 - r_i - dimreduce
#include "nrm_pvalue.h"

// per-grouping record (doubles): [0] ns, [1] vx (0 -> 1 applied), [2..25] p-value plan, then ccx (nc), then M^+ (nc*nc)
#define S1_HEAD 26

template <typename OutT>
__global__ void __launch_bounds__(256) k_s1_sweep(const double* __restrict__ G, int64_t ldg, const double* __restrict__ G2, int64_t ldg2,
												  const double* __restrict__ info, int64_t info_pitch, int nc, int64_t nx, int64_t ny,
												  int return_dot, OutT* __restrict__ p_out, OutT* __restrict__ stat_out,
												  OutT* __restrict__ vary_out, OutT* __restrict__ alpha_out, int64_t ldo,
												  int32_t* __restrict__ flags) {
	const int64_t i = blockIdx.y;
	const int64_t y = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (y >= ny) return;
	const double* rec = info + i * info_pitch;
	const double ns = rec[0], vx = rec[1];
	PvalPlan pl;
	pl.a = rec[2];
	pl.alpha = rec[3];
	pl.ln_front = rec[4];
	pl.umax = rec[5];
#pragma unroll
	for (int j = 0; j < NRM_PCOEF; j++) pl.coef[j] = rec[6 + j];
	const double* ccx = rec + S1_HEAD;
	const double* mi = ccx + nc;
	const double* g = G + y * ldg + i * (nc + 1);
	double ady = 0.0, adx = 0.0;  // a.ccy, a.ccx
	for (int c = 0; c < nc; c++) {
		double ccy = 0.0;
		for (int e = 0; e < nc; e++) ccy = fma(mi[c * nc + e], g[e], ccy);
		ady = fma(g[c], ccy, ady);
		adx = fma(g[c], ccx[c], adx);
	}
	const double yy = G2[y * ldg2 + i] - ady;
	const double xy = g[nc] - adx;
	const double vy = yy / ns;
	const double gam = xy / (ns * vx);
	const double r2 = gam * gam * vx / vy;
	if (flags) {
		if (!isfinite(r2) || !isfinite(vy)) atomicAdd(&flags[0], 1);
		else if (r2 > 1.0 + 1e-8) atomicAdd(&flags[1], 1);
	}
	const int64_t o = i * ldo + y;
	p_out[o] = (OutT)nrm_pvalue(r2, pl);
	stat_out[o] = (OutT)(return_dot ? gam * vx : gam);
	vary_out[o] = (OutT)vy;
	if (alpha_out) {
		for (int c = 0; c < nc; c++) {
			double ccy = 0.0;
			for (int e = 0; e < nc; e++) ccy = fma(mi[c * nc + e], g[e], ccy);
			alpha_out[o * nc + c] = (OutT)(ccy - gam * ccx[c]);  // association.py:368-370
		}
	}
}

extern "C" int nrm_single1_sweep(const double* d_g, int64_t ldg, const double* d_g2, int64_t ldg2, const double* d_info, int64_t info_pitch,
								 int64_t nc, int64_t nx, int64_t ny, int return_dot, void* d_p, void* d_stat, void* d_vary, void* d_alpha,
								 int out_dtype, int64_t ldo, int32_t* d_flags, void* stream) {
	NRM_REQUIRE(nx > 0 && ny > 0 && nc >= 0, "nrm_single1_sweep: bad sizes");
	NRM_REQUIRE(info_pitch >= S1_HEAD + nc + nc * nc, "nrm_single1_sweep: info pitch too small");
	NRM_REQUIRE(ldg >= nx * (nc + 1) && ldg2 >= nx && ldo >= ny, "nrm_single1_sweep: pitch too small");
	NRM_REQUIRE(out_dtype == NRM_F32 || out_dtype == NRM_F64, "nrm_single1_sweep: bad out_dtype");
	NRM_REQUIRE(d_g && d_g2 && d_info && d_p && d_stat && d_vary, "nrm_single1_sweep: null pointer");
	dim3 grid((unsigned)((ny + 255) / 256), (unsigned)nx);
	if (out_dtype == NRM_F64)
		hipLaunchKernelGGL(k_s1_sweep<double>, grid, dim3(256), 0, (hipStream_t)stream, d_g, ldg, d_g2, ldg2, d_info, info_pitch, (int)nc, nx, ny,
						   return_dot, (double*)d_p, (double*)d_stat, (double*)d_vary, (double*)d_alpha, ldo, d_flags);
	else
		hipLaunchKernelGGL(k_s1_sweep<float>, grid, dim3(256), 0, (hipStream_t)stream, d_g, ldg, d_g2, ldg2, d_info, info_pitch, (int)nc, nx, ny,
						   return_dot, (float*)d_p, (float*)d_stat, (float*)d_vary, (float*)d_alpha, ldo, d_flags);
	return nrm_check_launch("k_s1_sweep");
}

// ---- the same statistics without the masked Gram contractions, for designs whose entries are >= 0 (gRNA incidence) -----------------
// Then "cell k carries no OTHER grouping than i" (association.py:915-916) means: every other row of dx is 0 at k.  S_i is the union of
// N (cells where ALL of dx is 0: the same for every grouping) and E_i (cells where only row i is not 0), so every sum over S_i is a sum
// over N -- computed ONCE for every gene -- plus a sum over the few dozen cells of E_i, which k_s1_cells takes inside the sweep itself.
// 2 ny (nc + 2) (|N| + sum_i |E_i|) flop instead of 2 ny nx (nc + 2) n for the masked Gram: 1000 groupings cost what one costs.
//
// Round 4: the expression matrix is read ONCE, where it lies and in its own layout (k_s1_stream: a workgroup per S1_R gene rows streams
// them with 16-byte loads; the sums over N stay in registers, the values at the cells of the E_i -- a third of the matrix at one gRNA
// per cell -- leave in the order of the groupings, transposed, so that the sweep reads them coalesced with a thread per gene).  Before,
// torch gathered the permuted columns and transposed the result (two more passes over the matrix: 7.2 ms of the 9.9 ms of device time
// at 15 000 genes x 50 000 cells).
#define S1_NCMAX 32
#define S1_NCB 8  // covariates per pass of the stream kernel (more: further passes over the expression rows)

template <typename T>
struct S1Vec;
template <>
struct S1Vec<float> {
	static constexpr int V = 4;
	typedef float4 vec;
};
template <>
struct S1Vec<double> {
	static constexpr int V = 2;
	typedef double2 vec;
};

// V consecutive elements: one 16-byte load when the row is aligned, element loads otherwise (a template switch, not a branch: loads
// that meet behind a branch are issued one after the other)
template <typename T, int V, bool ALIGNED>
__device__ __forceinline__ void s1_ld(const T* __restrict__ p, T (&v)[V]) {
	if constexpr (ALIGNED) {
		typedef T vt __attribute__((ext_vector_type(V)));
		const vt t = *reinterpret_cast<const vt*>(p);
#pragma unroll
		for (int j = 0; j < V; j++) v[j] = t[j];
	} else {
#pragma unroll
		for (int j = 0; j < V; j++) v[j] = p[j];
	}
}

// cell codes: >= 0 position of the cell among the cells of the E_i (ordered by grouping), S1_COMMON a cell of N, S1_SKIP neither
#define S1_COMMON (-2)
#define S1_SKIP (-1)

// One workgroup = R gene rows, every cell.  common[(comp0 + c) * ny + y] = sum over N of y C_c (c < NC); FIRST: common[qrow * ny + y] =
// sum over N of y^2 and YE[pos][y] = y at the cell of position pos.  Rows past ny repeat row ny - 1 (their sums are not stored; their
// values land in the padding columns of YE, ldye >= ny rounded up to 8).
// (Round 6, measured and not kept: the NEXT group's rows and codes requested before this group's kept values are stored -- loads and stores share one
// in-order memory counter, and the ISA shows the wave waiting for its stores before the next iteration's loads are even issued -- 218 registers, no spill,
// and SLOWER: 1.60-1.65 ms against 1.22-1.33 in three alternating runs, profiles/r06_s1_stream_prefetch.txt.  The same kernel writing into another
// allocation of the same process: 1.20 against 1.25; a pitch of a whole 128-byte line instead of 8 values: no change.)
// (Measured and not kept: YE in blocks of R rows with the positions in the order of the cells, so that a wave's stores are one
// contiguous piece -- this kernel 1.28 -> 1.06 ms, but the sweep then gathers 32-byte pieces: 0.35 -> 0.78 ms, and the sweep is the
// one the host waits for.  The four waves of a workgroup on R rows each, same cells at the same time (their four pieces fill a line,
// covariates and codes shared through L1): 1.41 ms with 5 covariates, 1.16 with none -- against 1.24 and 1.40.)
template <typename T, int NC, int R, bool ALIGNED, bool FIRST>
__global__ void __launch_bounds__(256, 2) k_s1_stream(const T* __restrict__ Y, int64_t ldy, const double* __restrict__ C, int64_t ldc,
													   const int32_t* __restrict__ code, int64_t n, int64_t ny, double* __restrict__ common,
													   int comp0, int qrow, T* __restrict__ YE, int64_t ldye) {
	constexpr int V = S1Vec<T>::V;
	const int tid = threadIdx.x;
	// Workgroups are dealt to the 8 XCDs in turn; blocks of rows that are neighbours in YE (their R values of a cell share a 128-byte
	// line) go to the SAME XCD one after the other, so that the pieces of a line meet in one L2 before it is written back (1.50 -> 1.28 ms).
	const int64_t per = (gridDim.x + 7) / 8;
	const int64_t blk = (int64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
	const int64_t y0 = blk * R;
	if (y0 >= ny) return;
	const T* row[R];
#pragma unroll
	for (int r = 0; r < R; r++) row[r] = Y + (y0 + r < ny ? y0 + r : ny - 1) * ldy;
	double a[R][NC > 0 ? NC : 1], q[R];
#pragma unroll
	for (int r = 0; r < R; r++) {
		q[r] = 0.0;
#pragma unroll
		for (int c = 0; c < NC; c++) a[r][c] = 0.0;
	}
	auto take = [&](const int32_t(&cd)[V], const double(&cv)[NC > 0 ? NC : 1][V], const T(&yv)[R][V], int lanes) {
#pragma unroll
		for (int v = 0; v < V; v++) {
			if (v >= lanes) break;
			const bool m = cd[v] == S1_COMMON;
#pragma unroll
			for (int r = 0; r < R; r++) {
				const double yd = (double)yv[r][v];
				const double ym = m ? yd : 0.0;
				if constexpr (FIRST) q[r] = fma(ym, yd, q[r]);
#pragma unroll
				for (int c = 0; c < NC; c++) a[r][c] = fma(ym, cv[c][v], a[r][c]);
			}
			if constexpr (FIRST) {
				if (cd[v] >= 0) {
					typedef T st __attribute__((ext_vector_type(R)));
					st o;
#pragma unroll
					for (int r = 0; r < R; r++) o[r] = yv[r][v];
					*reinterpret_cast<st*>(YE + (int64_t)cd[v] * ldye + y0) = o;  // R values: R * sizeof(T) bytes, aligned (ldye % 8 == 0)
				}
			}
		}
	};
	const int64_t groups = n / V;
	for (int64_t g = tid; g < groups; g += 256) {
		const int64_t k = g * V;
		int32_t cd[V];
		double cv[NC > 0 ? NC : 1][V];
		T yv[R][V];
		s1_ld<int32_t, V, ALIGNED>(code + k, cd);
#pragma unroll
		for (int c = 0; c < NC; c++) {
			if constexpr (ALIGNED) {
#pragma unroll
				for (int h = 0; h < V; h += 2) {
					const double2 t = *reinterpret_cast<const double2*>(C + c * ldc + k + h);
					cv[c][h] = t.x;
					cv[c][h + 1] = t.y;
				}
			} else {
#pragma unroll
				for (int h = 0; h < V; h++) cv[c][h] = C[c * ldc + k + h];
			}
		}
#pragma unroll
		for (int r = 0; r < R; r++) s1_ld<T, V, ALIGNED>(row[r] + k, yv[r]);
		take(cd, cv, yv, V);
	}
	{  // the last n % V cells: one cell each for the first few threads
		const int64_t k = groups * V + tid;
		if (k < n) {
			int32_t cd[V];
			double cv[NC > 0 ? NC : 1][V];
			T yv[R][V];
			cd[0] = code[k];
#pragma unroll
			for (int c = 0; c < NC; c++) cv[c][0] = C[c * ldc + k];
#pragma unroll
			for (int r = 0; r < R; r++) yv[r][0] = row[r][k];
			take(cd, cv, yv, 1);
		}
	}
	// sums of the 256 threads: within a wave, then the four waves in order
	constexpr int NV = R * (NC + (FIRST ? 1 : 0));
	__shared__ double red[4][NV > 0 ? NV : 1];
	const int lane = tid & 63, w = tid >> 6;
#pragma unroll
	for (int r = 0; r < R; r++) {
#pragma unroll
		for (int c = 0; c < NC; c++) {
			double t = a[r][c];
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
			if (lane == 0) red[w][r * NC + c] = t;
		}
		if constexpr (FIRST) {
			double t = q[r];
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
			if (lane == 0) red[w][R * NC + r] = t;
		}
	}
	__syncthreads();
	if (tid < NV) {
		const double t = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
		const int r = tid < R * NC ? tid / (NC > 0 ? NC : 1) : tid - R * NC;
		const int comp = tid < R * NC ? comp0 + tid % (NC > 0 ? NC : 1) : qrow;
		if (y0 + r < ny) common[(int64_t)comp * ny + y0 + r] = t;
	}
}

// The sweep: thread (grouping i, gene y) adds the cells of E_i to the sums over N and finishes the pair.
//   YE (cells of the E_i, ldye) expression values, CE (cells, nc) fp64 covariates, xe (cells) the grouping's own value, seg[i] .. seg[i+1]
//   the cells of grouping i; common[c * ny + y] (c < nc), common[qrow * ny + y] from k_s1_stream.
// NCT: the number of covariates when it is at most 8 (loops unrolled, sums in registers without predicates: the generic form spends
// its time on 32 compare-and-branch pairs per cell), -1: any number up to S1_NCMAX.
template <typename OutT>
__device__ __forceinline__ void s1_put(void* base, int64_t o, double v) {
	reinterpret_cast<OutT*>(base)[o] = (OutT)v;
}

template <typename T, int NCT>
__global__ void __launch_bounds__(256) k_s1_cells(const T* __restrict__ YE, int64_t ldye, const double* __restrict__ CE, const double* __restrict__ xe,
												   const int64_t* __restrict__ seg, const double* __restrict__ common, int qrow,
												   const double* __restrict__ info, int64_t info_pitch, int nc_rt, int64_t nx, int64_t ny, int return_dot,
												   void* __restrict__ p_out, void* __restrict__ stat_out, void* __restrict__ vary_out,
												   void* __restrict__ alpha_out, int out_f64, int64_t ldo, int32_t* __restrict__ flags) {
	constexpr int NA = NCT >= 0 ? (NCT > 0 ? NCT : 1) : S1_NCMAX;
	const int nc = NCT >= 0 ? NCT : nc_rt;
	const int64_t i = blockIdx.y;
	const int64_t y = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (y >= ny) return;
	const double* rec = info + i * info_pitch;
	const double ns = rec[0], vx = rec[1];
	PvalPlan pl;
	pl.a = rec[2];
	pl.alpha = rec[3];
	pl.ln_front = rec[4];
	pl.umax = rec[5];
#pragma unroll
	for (int j = 0; j < NRM_PCOEF; j++) pl.coef[j] = rec[6 + j];
	const double* ccx = rec + S1_HEAD;
	const double* mi = ccx + nc;
	// a = y C_S^T, xy = y . x_S, q = |y_S|^2: the common part (x is 0 on N) plus this grouping's own cells
	double a[NA], xy = 0.0, q = common[(int64_t)qrow * ny + y];
#pragma unroll
	for (int c = 0; c < NA; c++) a[c] = c < nc ? common[(int64_t)c * ny + y] : 0.0;
	int64_t k = seg[i];
	const int64_t k1 = seg[i + 1];
#ifndef S1_CU
#define S1_CU 8  // (4: 0.347 ms, 8: 0.333, 16: 0.359 for the configs[3]-shaped screen)
#endif
	for (; k + S1_CU <= k1; k += S1_CU) {  // S1_CU cells per step: their loads are in flight together
		double v[S1_CU];
#pragma unroll
		for (int u = 0; u < S1_CU; u++) v[u] = (double)YE[(k + u) * ldye + y];
#pragma unroll
		for (int u = 0; u < S1_CU; u++) {
			const double* ck = CE + (k + u) * nc;
#pragma unroll
			for (int c = 0; c < NA; c++)
				if (c < nc) a[c] = fma(v[u], ck[c], a[c]);
			xy = fma(v[u], xe[k + u], xy);
			q = fma(v[u], v[u], q);
		}
	}
	for (; k < k1; k++) {
		const double v = (double)YE[k * ldye + y];
		const double* ck = CE + k * nc;
#pragma unroll
		for (int c = 0; c < NA; c++)
			if (c < nc) a[c] = fma(v, ck[c], a[c]);
		xy = fma(v, xe[k], xy);
		q = fma(v, v, q);
	}
	double ady = 0.0, adx = 0.0;  // a.ccy, a.ccx
	double ccy_[NCT > 0 ? NCT : 1];
	if constexpr (NCT >= 0) {
#pragma unroll
		for (int c = 0; c < NCT; c++) {
			double ccy = 0.0;
#pragma unroll
			for (int e = 0; e < NCT; e++) ccy = fma(mi[c * NCT + e], a[e], ccy);
			ccy_[c] = ccy;
			ady = fma(a[c], ccy, ady);
			adx = fma(a[c], ccx[c], adx);
		}
	} else {
		for (int c = 0; c < nc; c++) {
			double ccy = 0.0;
#pragma unroll
			for (int e = 0; e < NA; e++)
				if (e < nc) ccy = fma(mi[c * nc + e], a[e], ccy);
			double ac = 0.0;
#pragma unroll
			for (int e = 0; e < NA; e++)
				if (e == c) ac = a[e];
			ady = fma(ac, ccy, ady);
			adx = fma(ac, ccx[c], adx);
		}
	}
	const double yy = q - ady;
	xy -= adx;
	const double vy = yy / ns;
	const double gam = xy / (ns * vx);
	const double r2 = gam * gam * vx / vy;
	if (flags) {
		if (!isfinite(r2) || !isfinite(vy)) atomicAdd(&flags[0], 1);
		else if (r2 > 1.0 + 1e-8) atomicAdd(&flags[1], 1);
	}
	const int64_t o = i * ldo + y;
	const double pv = nrm_pvalue(r2, pl), st = return_dot ? gam * vx : gam;
	if (out_f64) {
		s1_put<double>(p_out, o, pv);
		s1_put<double>(stat_out, o, st);
		s1_put<double>(vary_out, o, vy);
	} else {
		s1_put<float>(p_out, o, pv);
		s1_put<float>(stat_out, o, st);
		s1_put<float>(vary_out, o, vy);
	}
	if (alpha_out) {
		for (int c = 0; c < nc; c++) {
			double ccy = 0.0;
			if constexpr (NCT >= 0) {
#pragma unroll
				for (int e = 0; e < NCT; e++)
					if (e == c) ccy = ccy_[e];
			} else {
#pragma unroll
				for (int e = 0; e < NA; e++)
					if (e < nc) ccy = fma(mi[c * nc + e], a[e], ccy);
			}
			const double al = ccy - gam * ccx[c];  // association.py:368-370
			if (out_f64)
				s1_put<double>(alpha_out, o * nc + c, al);
			else
				s1_put<float>(alpha_out, o * nc + c, al);
		}
	}
}

template <typename T, int NC, int R, bool FIRST>
static void s1_stream_go(const void* d_y, int64_t ldy, const double* d_c, int64_t ldc, const int32_t* d_code, int64_t n, int64_t ny, double* d_common,
						 int comp0, int qrow, void* d_ye, int64_t ldye, hipStream_t st) {
	const bool aligned = ((uintptr_t)d_y % 16 == 0) && (ldy * sizeof(T)) % 16 == 0 && ((uintptr_t)d_code % 16 == 0) &&
						 (NC == 0 || (((uintptr_t)d_c % 16 == 0) && ldc % 2 == 0));
	const dim3 grid((unsigned)(((ny + R - 1) / R + 7) / 8 * 8));
	if (aligned)
		hipLaunchKernelGGL((k_s1_stream<T, NC, R, true, FIRST>), grid, dim3(256), 0, st, (const T*)d_y, ldy, d_c, ldc, d_code, n, ny, d_common, comp0, qrow,
						   (T*)d_ye, ldye);
	else
		hipLaunchKernelGGL((k_s1_stream<T, NC, R, false, FIRST>), grid, dim3(256), 0, st, (const T*)d_y, ldy, d_c, ldc, d_code, n, ny, d_common, comp0, qrow,
						   (T*)d_ye, ldye);
}

template <typename T>
static int s1_stream(const void* d_y, int64_t ldy, const double* d_c, int64_t ldc, int64_t nc, const int32_t* d_code, int64_t n, int64_t ny,
					 double* d_common, void* d_ye, int64_t ldye, hipStream_t st) {
	const int qrow = (int)nc;
#define S1_FIRST(NC, R)                                                                                           \
	case NC:                                                                                                      \
		s1_stream_go<T, NC, R, true>(d_y, ldy, d_c, ldc, d_code, n, ny, d_common, 0, qrow, d_ye, ldye, st); \
		break;
	const int first = (int)(nc <= S1_NCB ? nc : S1_NCB);
	switch (first) {
		S1_FIRST(0, 8)
		S1_FIRST(1, 8)
		S1_FIRST(2, 8)
		S1_FIRST(3, 8)
		S1_FIRST(4, 8)
		S1_FIRST(5, 8)
		S1_FIRST(6, 4)
		S1_FIRST(7, 4)
		S1_FIRST(8, 4)
	}
#undef S1_FIRST
	// further covariates, up to eight per pass over the expression rows (the last pass may reach back over covariates already done:
	// the same sums, stored twice)
	for (int64_t c0 = S1_NCB; c0 < nc; c0 += S1_NCB) {
		const int64_t b = c0 + S1_NCB <= nc ? c0 : nc - S1_NCB;
		s1_stream_go<T, S1_NCB, 4, false>(d_y, ldy, d_c + b * ldc, ldc, d_code, n, ny, d_common, (int)b, qrow, nullptr, ldye, st);
	}
	return nrm_check_launch("k_s1_stream");
}

extern "C" int nrm_single1_stream(const void* d_y, int y_dtype, int64_t ldy, const double* d_c, int64_t ldc, int64_t nc, const int32_t* d_code,
								  int64_t n, int64_t ny, double* d_common, void* d_ye, int64_t ldye, void* stream) {
	NRM_REQUIRE(n > 0 && ny > 0 && nc >= 0 && nc <= S1_NCMAX, "nrm_single1_stream: bad sizes (at most %d covariates)", S1_NCMAX);
	NRM_REQUIRE(y_dtype == NRM_F32 || y_dtype == NRM_F64, "nrm_single1_stream: bad dtype");
	NRM_REQUIRE(ldy >= n && (nc == 0 || ldc >= n) && ldye >= ny && ldye % 8 == 0, "nrm_single1_stream: bad pitch (ldye: a multiple of 8, at least ny)");
	NRM_REQUIRE(d_y && d_code && d_common && d_ye && (d_c || nc == 0), "nrm_single1_stream: null pointer");
	NRM_REQUIRE((uintptr_t)d_ye % 64 == 0, "nrm_single1_stream: d_ye must be 64-byte aligned");
	if (y_dtype == NRM_F64) return s1_stream<double>(d_y, ldy, d_c, ldc, nc, d_code, n, ny, d_common, d_ye, ldye, (hipStream_t)stream);
	return s1_stream<float>(d_y, ldy, d_c, ldc, nc, d_code, n, ny, d_common, d_ye, ldye, (hipStream_t)stream);
}

extern "C" int nrm_single1_cells(const void* d_ye, int y_dtype, int64_t ldye, const double* d_ce, const double* d_xe, const int64_t* d_seg,
								 const double* d_common, const double* d_info, int64_t info_pitch, int64_t nc, int64_t nx, int64_t ny, int return_dot,
								 void* d_p, void* d_stat, void* d_vary, void* d_alpha, int out_dtype, int64_t ldo, int32_t* d_flags, void* stream) {
	NRM_REQUIRE(nx > 0 && ny > 0 && nc >= 0 && nc <= S1_NCMAX, "nrm_single1_cells: bad sizes (at most %d covariates)", S1_NCMAX);
	NRM_REQUIRE(info_pitch >= S1_HEAD + nc + nc * nc && ldo >= ny && ldye >= ny, "nrm_single1_cells: pitch too small");
	NRM_REQUIRE((y_dtype == NRM_F32 || y_dtype == NRM_F64) && (out_dtype == NRM_F32 || out_dtype == NRM_F64), "nrm_single1_cells: bad dtype");
	NRM_REQUIRE(d_ye && d_xe && d_seg && d_common && d_info && d_p && d_stat && d_vary && (d_ce || nc == 0), "nrm_single1_cells: null pointer");
	hipStream_t st = (hipStream_t)stream;
	const dim3 grid((unsigned)((ny + 255) / 256), (unsigned)nx);
	const int f64 = out_dtype == NRM_F64;
#define S1_GO(T, NCT)                                                                                                                           \
	hipLaunchKernelGGL((k_s1_cells<T, NCT>), grid, dim3(256), 0, st, (const T*)d_ye, ldye, d_ce, d_xe, d_seg, d_common, (int)nc, d_info, info_pitch, \
					   (int)nc, nx, ny, return_dot, d_p, d_stat, d_vary, d_alpha, f64, ldo, d_flags)
#define S1_CASE(NCT)               \
	case NCT:                      \
		if (y_dtype == NRM_F64)    \
			S1_GO(double, NCT);    \
		else                       \
			S1_GO(float, NCT);     \
		break;
	switch (nc <= 8 ? (int)nc : -1) {
		S1_CASE(0)
		S1_CASE(1)
		S1_CASE(2)
		S1_CASE(3)
		S1_CASE(4)
		S1_CASE(5)
		S1_CASE(6)
		S1_CASE(7)
		S1_CASE(8)
		S1_CASE(-1)
	}
#undef S1_CASE
#undef S1_GO
	return nrm_check_launch("k_s1_cells");
}

// ---- the groupings' own statistics over their own cells ------------------------------------------------------------------------------------
// M_i = C_S C_S^T, C_S x_S and |x_S|^2 (association.py:350-364) are sums over the shared cells -- the same for every grouping, taken once
// -- plus sums over the few dozen cells of E_i: a wave per grouping walks them (lane-strided, then a tree over the lanes: a fixed order).
// out (nx, np): the nc (nc + 1) / 2 products C_c C_d (c <= d, row by row), then C_c x (nc), then x x.  The host did this with numpy
// segment sums: 1.1 ms of the 3.9 ms of a call at BASELINE configs[3] size, on its critical path.
#define S1_GS_NC 8
__global__ void __launch_bounds__(64) k_s1_group_stats(const int64_t* __restrict__ seg, const int64_t* __restrict__ cells, const double* __restrict__ xe,
														const double* __restrict__ C, int64_t ldc, int nc, int64_t nx, double* __restrict__ out) {
	constexpr int NPMAX = S1_GS_NC * (S1_GS_NC + 1) / 2 + S1_GS_NC + 1;
	const int64_t i = blockIdx.x;
	const int lane = threadIdx.x;
	double acc[NPMAX];
#pragma unroll
	for (int j = 0; j < NPMAX; j++) acc[j] = 0.0;
	for (int64_t e = seg[i] + lane; e < seg[i + 1]; e += 64) {
		const int64_t k = cells[e];
		const double x = xe[e];
		double cv[S1_GS_NC];
#pragma unroll
		for (int c = 0; c < S1_GS_NC; c++) cv[c] = c < nc ? C[c * ldc + k] : 0.0;
		int j = 0;
#pragma unroll
		for (int c = 0; c < S1_GS_NC; c++)
#pragma unroll
			for (int d = c; d < S1_GS_NC; d++, j++) acc[j] = fma(cv[c], cv[d], acc[j]);
#pragma unroll
		for (int c = 0; c < S1_GS_NC; c++, j++) acc[j] = fma(cv[c], x, acc[j]);
		acc[j] = fma(x, x, acc[j]);
	}
#pragma unroll
	for (int j = 0; j < NPMAX; j++) {
		double t = acc[j];
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
		acc[j] = t;
	}
	if (lane == 0) {
		// the static tables hold S1_GS_NC covariates; the caller's layout is for nc of them
		double* o = out + i * (nc * (nc + 1) / 2 + nc + 1);
		int j = 0, w = 0;
#pragma unroll
		for (int c = 0; c < S1_GS_NC; c++)
#pragma unroll
			for (int d = c; d < S1_GS_NC; d++, j++)
				if (c < nc && d < nc) o[w++] = acc[j];
#pragma unroll
		for (int c = 0; c < S1_GS_NC; c++, j++)
			if (c < nc) o[w++] = acc[j];
		o[w] = acc[j];
	}
}

// d_seg (nx + 1): the own cells of grouping i are entries [d_seg[i], d_seg[i + 1]) of d_cells (cell indices) / d_xe (the grouping's value there);
// d_out (nx, nc (nc + 1) / 2 + nc + 1) as above.  nc <= 8.
extern "C" int nrm_single1_group_stats(const int64_t* d_seg, const int64_t* d_cells, const double* d_xe, const double* d_c, int64_t ldc, int64_t nc, int64_t nx,
									   double* d_out, void* stream) {
	NRM_REQUIRE(nx > 0 && nc >= 0 && nc <= S1_GS_NC && d_seg && d_cells && d_xe && d_out && (nc == 0 || d_c), "nrm_single1_group_stats: bad arguments (at most %d covariates)",
				S1_GS_NC);
	hipLaunchKernelGGL(k_s1_group_stats, dim3((unsigned)nx), dim3(64), 0, (hipStream_t)stream, d_seg, d_cells, d_xe, d_c, ldc, (int)nc, nx, d_out);
	return nrm_check_launch("k_s1_group_stats");
}

// ---- per grouping: pseudo-inverse, rank, ccx, vx, dof and the P-value plan -- the record k_s1_cells reads -- without the host ------------------
// Rounds 3-5 finished the groupings' statistics on the host (1000 small pseudo-inverses, two einsum calls, 1000 P-value plans, five read-backs and
// an upload per call: "a single=1 step is bound by the host", and on a box whose host was slow the 2 ms step took 18).  A lane per grouping does
// the same here, with the same code where integers are decided (nrm_jacobi.h: the rank rule of association.py:77-80) --
//     M_i = C_N C_N^T + C_Ei C_Ei^T (the shared cells' partial sums added in block order + the grouping's own, k_s1_group_stats)
//     M_i^+, r_i;  ccx_i = M_i^+ (C_S x_S);  vx_i = (|x_S|^2 - (C_S x_S) . ccx_i) / ns_i, 0 -> 1        association.py:350-364
//     dof_i = ns_i - 1 - r_i - dimreduce and its P-value plan (nrm_pvalue_plan.h)                         association.py:372-374
// and what the reference asserts or raises on the way is COUNTED into flags (the host looks once, when it takes the results):
//     flags[2] groupings with a single value on their selected cells (association.py:917-918, AssertionError)
//     flags[3] groupings with dof <= 0 ("Insufficient number of cells")        flags[4] groupings whose M_i is not finite (the SVD's ValueError)
#include "nrm_jacobi.h"
#include "nrm_pvalue_plan.h"

template <int NC>
__global__ void __launch_bounds__(64) k_s1_group_info(const double* __restrict__ gs, const double* __restrict__ gpart, int gb, const double* __restrict__ rowinfo,
													   const int64_t* __restrict__ sel_info, int64_t nx, int dimreduce, double* __restrict__ info, int64_t pitch,
													   double* __restrict__ varx, int32_t* __restrict__ flags) {
	constexpr int NA = NC > 0 ? NC : 1, NP = NC * (NC + 1) / 2;
	__shared__ double mcc[64];
	if constexpr (NC > 0) {
		double t = 0.0;
		for (int g = 0; g < gb; g++) t += gpart[(int64_t)g * 64 + threadIdx.x];  // (block order: the sum the host took)
		mcc[threadIdx.x] = t;
		__syncthreads();
	}
	const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
	if (i >= nx) return;
	const double n_common = (double)sel_info[3];
	const double ns = n_common + rowinfo[i * 3];
	double lo = rowinfo[i * 3 + 1], hi = rowinfo[i * 3 + 2];
	if (n_common > 0) {  // 0 on the shared cells, if there are any, and its own values on its own
		lo = fmin(lo, 0.0);
		hi = fmax(hi, 0.0);
	}
	if (!(hi > lo)) atomicAdd(&flags[2], 1);
	const double* o = gs + i * (NP + NC + 1);
	double* rec = info + i * pitch;
	double xx = o[NP + NC];
	int64_t rk = 0;
	if constexpr (NC > 0) {
		double m[NA * NA], inv[NA * NA];
		bool finite = true;
		int w = 0;
		for (int c = 0; c < NC; c++)
			for (int d = c; d < NC; d++, w++) {
				const double v = o[w] + mcc[c * 8 + d];
				m[c * NC + d] = m[d * NC + c] = v;
				finite = finite && isfinite(v);
			}
		if (!finite) {
			atomicAdd(&flags[4], 1);
			for (int e = 0; e < NC * NC; e++) m[e] = e % (NC + 1) == 0 ? 1.0 : 0.0;
		}
		nrm_small_pinv_one<NA>(m, NC, 1e-8, inv, &rk);  // association.py:350-351
		if (rk == 0)
			for (int e = 0; e < NC * NC; e++) inv[e] = 0.0;
		const double* xc = o + NP;
		double dot = 0.0;
		for (int c = 0; c < NC; c++) {
			double t = 0.0;
			for (int d = 0; d < NC; d++) t += inv[c * NC + d] * xc[d];
			rec[S1_HEAD + c] = t;  // ccx
			dot += xc[c] * t;
		}
		xx -= dot;
		for (int e = 0; e < NC * NC; e++) rec[S1_HEAD + NC + e] = inv[e];
	}
	double vx = xx / ns;
	if (vx == 0.0) vx = 1.0;  // association.py:362-364
	double dof = ns - 1.0 - (double)rk - (double)dimreduce;
	if (!(dof > 0.0)) {
		atomicAdd(&flags[3], 1);
		dof = 1.0;  // (the call raises; the sweep still gets a plan it can evaluate)
	}
	rec[0] = ns;
	rec[1] = vx;
	nrm_pvalue_plan_fill(dof, rec + 2);
	varx[i] = vx;
}

// d_gs (nx, nc (nc + 1) / 2 + nc + 1) from nrm_single1_group_stats; d_gram_part, d_rowinfo, d_sel_info from nrm_single1_select (d_gram_part may be NULL
// for nc == 0); d_info (nx, info_pitch) <- the records nrm_single1_cells reads; d_varx (nx) fp64; d_flags int32[8] (see above; [0], [1] are the sweep's).
// nc <= 8.  Reference: association.py:350-374 (the loop body of association_test_2 on the grouping's side).
extern "C" int nrm_single1_group_info(const double* d_gs, const double* d_gram_part, const double* d_rowinfo, const int64_t* d_sel_info, int64_t nc, int64_t nx,
									  int dimreduce, double* d_info, int64_t info_pitch, double* d_varx, int32_t* d_flags, void* stream) {
	NRM_REQUIRE(nx > 0 && nc >= 0 && nc <= S1_GS_NC && info_pitch >= S1_HEAD + nc + nc * nc, "nrm_single1_group_info: bad sizes (at most %d covariates)", S1_GS_NC);
	NRM_REQUIRE(d_gs && d_rowinfo && d_sel_info && d_info && d_varx && d_flags && (nc == 0 || d_gram_part), "nrm_single1_group_info: null pointer");
	const dim3 grid((unsigned)((nx + 63) / 64));
	const int gb = (int)nrm_single1_select_gram_blocks();
#define S1_GI(NCV)                                                                                                                                                  \
	case NCV:                                                                                                                                                       \
		hipLaunchKernelGGL((k_s1_group_info<NCV>), grid, dim3(64), 0, (hipStream_t)stream, d_gs, d_gram_part, gb, d_rowinfo, d_sel_info, nx, dimreduce, d_info, info_pitch, \
						   d_varx, d_flags);                                                                                                                        \
		break;
	switch ((int)nc) {
		S1_GI(0) S1_GI(1) S1_GI(2) S1_GI(3) S1_GI(4) S1_GI(5) S1_GI(6) S1_GI(7) S1_GI(8)
	}
#undef S1_GI
	return nrm_check_launch("k_s1_group_info");
}

// The double-precision plans of nrm_pvalue_plan.h on the host (the code the device runs, compiled for the host: tests hold it to nrm_pvalue_plan_init_many)
extern "C" int nrm_pvalue_plan_fill_many(const double* dof, int64_t count, double* out, int64_t pitch) {
	NRM_REQUIRE(count >= 0 && pitch >= 4 + NRM_PCOEF && (count == 0 || (dof && out)), "nrm_pvalue_plan_fill_many: bad arguments");
	for (int64_t j = 0; j < count; j++) {
		NRM_REQUIRE(dof[j] > 0, "Insufficient number of cells: dof = %g must be positive", dof[j]);
		nrm_pvalue_plan_fill(dof[j], out + j * pitch);
	}
	return NRM_OK;
}
