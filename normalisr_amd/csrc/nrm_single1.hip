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
// over N -- computed ONCE for every gene (k_s1_common) -- plus a sum over the few dozen cells of E_i, which k_s1_sparse takes inside
// the sweep itself.  2 ny (nc + 2) (|N| + sum_i |E_i|) flop instead of 2 ny nx (nc + 2) n for the masked Gram: 1000 groupings cost
// what one costs.  The cells are permuted (N first, then the E_i one after another) and the expression matrix is handed over
// transposed in that order (YT: cells x genes), so that a thread per gene reads coalesced.
#define S1_NCMAX 32
#define S1_PIECE 512

// partial sums over piece blockIdx.y of the N cells: part[piece][y][0 .. nc) = sum y C_c, [nc] = sum y^2
template <typename T>
__global__ void __launch_bounds__(256) k_s1_common(const T* __restrict__ YT, int64_t ldy, const double* __restrict__ CT, int nc, int64_t n_common,
													int64_t ny, double* __restrict__ part) {
	const int64_t y = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (y >= ny) return;
	const int64_t k0 = (int64_t)blockIdx.y * S1_PIECE, k1 = k0 + S1_PIECE < n_common ? k0 + S1_PIECE : n_common;
	double a[S1_NCMAX], q = 0.0;
#pragma unroll
	for (int c = 0; c < S1_NCMAX; c++) a[c] = 0.0;
	for (int64_t k = k0; k < k1; k++) {
		const double v = (double)YT[k * ldy + y];
		const double* ck = CT + k * nc;
#pragma unroll
		for (int c = 0; c < S1_NCMAX; c++)
			if (c < nc) a[c] = fma(v, ck[c], a[c]);
		q = fma(v, v, q);
	}
	double* o = part + ((int64_t)blockIdx.y * ny + y) * (nc + 1);
#pragma unroll
	for (int c = 0; c < S1_NCMAX; c++)
		if (c < nc) o[c] = a[c];
	o[nc] = q;
}

// common[y][0 .. nc] = the pieces added in order
__global__ void __launch_bounds__(256) k_s1_common_sum(const double* __restrict__ part, int pieces, int64_t ny, int nw, double* __restrict__ common) {
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= ny * nw) return;
	double acc = 0.0;
	for (int p = 0; p < pieces; p++) acc += part[(int64_t)p * ny * nw + e];
	common[e] = acc;
}

template <typename T, typename OutT>
__global__ void __launch_bounds__(256) k_s1_sparse(const T* __restrict__ YT, int64_t ldy, const double* __restrict__ CT, const double* __restrict__ xp,
													const int64_t* __restrict__ seg, const double* __restrict__ common, const double* __restrict__ info,
													int64_t info_pitch, int nc, int64_t nx, int64_t ny, int return_dot, OutT* __restrict__ p_out,
													OutT* __restrict__ stat_out, OutT* __restrict__ vary_out, OutT* __restrict__ alpha_out, int64_t ldo,
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
	// a = y C_S^T, xy = y . x_S, q = |y_S|^2: the common part (x is 0 on N) plus this grouping's own cells
	const double* cm = common + y * (nc + 1);
	double a[S1_NCMAX], xy = 0.0, q = cm[nc];
#pragma unroll
	for (int c = 0; c < S1_NCMAX; c++) a[c] = c < nc ? cm[c] : 0.0;
	for (int64_t k = seg[i]; k < seg[i + 1]; k++) {
		const double v = (double)YT[k * ldy + y];
		const double* ck = CT + k * nc;
#pragma unroll
		for (int c = 0; c < S1_NCMAX; c++)
			if (c < nc) a[c] = fma(v, ck[c], a[c]);
		xy = fma(v, xp[k], xy);
		q = fma(v, v, q);
	}
	double ady = 0.0, adx = 0.0;  // a.ccy, a.ccx
	for (int c = 0; c < nc; c++) {
		double ccy = 0.0;
#pragma unroll
		for (int e = 0; e < S1_NCMAX; e++)
			if (e < nc) ccy = fma(mi[c * nc + e], a[e], ccy);
		double ac = 0.0;
#pragma unroll
		for (int e = 0; e < S1_NCMAX; e++)
			if (e == c) ac = a[e];
		ady = fma(ac, ccy, ady);
		adx = fma(ac, ccx[c], adx);
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
	p_out[o] = (OutT)nrm_pvalue(r2, pl);
	stat_out[o] = (OutT)(return_dot ? gam * vx : gam);
	vary_out[o] = (OutT)vy;
	if (alpha_out) {
		for (int c = 0; c < nc; c++) {
			double ccy = 0.0;
#pragma unroll
			for (int e = 0; e < S1_NCMAX; e++)
				if (e < nc) ccy = fma(mi[c * nc + e], a[e], ccy);
			alpha_out[o * nc + c] = (OutT)(ccy - gam * ccx[c]);  // association.py:368-370
		}
	}
}

extern "C" int64_t nrm_single1_sparse_workspace_doubles(int64_t ny, int64_t nc, int64_t n_common) {
	return ny * (nc + 1) * ((n_common + S1_PIECE - 1) / S1_PIECE + 1);
}

extern "C" int nrm_single1_sparse(const void* d_yt, int y_dtype, int64_t ldy, const double* d_ct, const double* d_xp, const int64_t* d_seg,
								  int64_t n_common, const double* d_info, int64_t info_pitch, int64_t nc, int64_t nx, int64_t ny, int return_dot,
								  void* d_p, void* d_stat, void* d_vary, void* d_alpha, int out_dtype, int64_t ldo, double* d_work, int32_t* d_flags,
								  void* stream) {
	NRM_REQUIRE(nx > 0 && ny > 0 && nc >= 0 && nc <= S1_NCMAX && n_common >= 0, "nrm_single1_sparse: bad sizes (at most %d covariates)", S1_NCMAX);
	NRM_REQUIRE(info_pitch >= S1_HEAD + nc + nc * nc && ldo >= ny && ldy >= ny, "nrm_single1_sparse: pitch too small");
	NRM_REQUIRE((y_dtype == NRM_F32 || y_dtype == NRM_F64) && (out_dtype == NRM_F32 || out_dtype == NRM_F64), "nrm_single1_sparse: bad dtype");
	NRM_REQUIRE(d_yt && d_xp && d_seg && d_info && d_p && d_stat && d_vary && d_work && (d_ct || nc == 0), "nrm_single1_sparse: null pointer");
	hipStream_t st = (hipStream_t)stream;
	const int nw = (int)nc + 1;
	const int pieces = (int)((n_common + S1_PIECE - 1) / S1_PIECE);
	double* common = d_work;                 // (ny, nc + 1)
	double* part = d_work + ny * nw;         // (pieces, ny, nc + 1)
	const unsigned gy = (unsigned)((ny + 255) / 256);
	if (pieces > 0) {
		if (y_dtype == NRM_F64)
			hipLaunchKernelGGL(k_s1_common<double>, dim3(gy, (unsigned)pieces), dim3(256), 0, st, (const double*)d_yt, ldy, d_ct, (int)nc, n_common, ny, part);
		else
			hipLaunchKernelGGL(k_s1_common<float>, dim3(gy, (unsigned)pieces), dim3(256), 0, st, (const float*)d_yt, ldy, d_ct, (int)nc, n_common, ny, part);
	}
	hipLaunchKernelGGL(k_s1_common_sum, dim3((unsigned)((ny * nw + 255) / 256)), dim3(256), 0, st, part, pieces, ny, nw, common);
	const dim3 grid(gy, (unsigned)nx);
#define S1_GO(T, O)                                                                                                                         \
	hipLaunchKernelGGL((k_s1_sparse<T, O>), grid, dim3(256), 0, st, (const T*)d_yt, ldy, d_ct, d_xp, d_seg, common, d_info, info_pitch, (int)nc, nx, \
					   ny, return_dot, (O*)d_p, (O*)d_stat, (O*)d_vary, (O*)d_alpha, ldo, d_flags)
	if (y_dtype == NRM_F64) {
		if (out_dtype == NRM_F64)
			S1_GO(double, double);
		else
			S1_GO(double, float);
	} else {
		if (out_dtype == NRM_F64)
			S1_GO(float, double);
		else
			S1_GO(float, float);
	}
#undef S1_GO
	return nrm_check_launch("k_s1_sparse");
}
