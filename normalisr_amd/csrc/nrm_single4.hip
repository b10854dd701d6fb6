// single=4 ("other groupings as covariates", association.py:421-576,926-980) in closed form: one multiple
// regression of every gene on A = [dx; dc] replaces the reference's per-grouping SVD loop.
// The heavy contractions A A^T, Y A^T and (Y A^T) N run on K2; this file is the per-pair sweep.
#include "nrm_pvalue.h"

extern "C" int nrm_pvalue_plan_init(nrm_pvalue_plan* plan, double dof);

// rss[y] = yy[y] - sum_k Pt[y,k] Bt[y,k]   (residual sum of squares of the full regression); one wave per gene
__global__ void __launch_bounds__(256) k_s4_rss(const double* __restrict__ bt, const double* __restrict__ pt, int64_t ldb,
												const double* __restrict__ yy, int64_t ny, int64_t m, double* __restrict__ rss) {
	const int lane = threadIdx.x & 63;
	const int64_t y = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (y >= ny) return;
	double acc = 0.0;
	for (int64_t k = lane; k < m; k += 64) acc = fma(pt[y * ldb + k], bt[y * ldb + k], acc);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
	if (lane == 0) rss[y] = yy[y] - acc;
}

#define S4_T 64
template <typename OutT>
__global__ void __launch_bounds__(256) k_s4_sweep(const double* __restrict__ bt, int64_t ldb, const double* __restrict__ rss,
												  const double* __restrict__ dxx, int64_t nx, int64_t ny, double ncells,
												  int return_dot, PvalPlan pl, OutT* __restrict__ p_out, OutT* __restrict__ stat_out,
												  OutT* __restrict__ vary_out, int64_t ldo, int32_t* __restrict__ flags) {
	__shared__ double tile[S4_T][S4_T + 1];  // [gene][grouping]
	const int bi = blockIdx.y, bj = blockIdx.x;  // grouping block, gene block
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	for (int r = ty; r < S4_T; r += 4) {
		const int64_t gy = (int64_t)bj * S4_T + r, gi = (int64_t)bi * S4_T + tx;
		tile[r][tx] = (gy < ny && gi < nx) ? bt[gy * ldb + gi] : 0.0;
	}
	__syncthreads();
	const int64_t gy = (int64_t)bj * S4_T + tx;
	const double rs = gy < ny ? rss[gy] : 1.0;
	int bad_nf = 0, bad_rng = 0;
	for (int r = ty; r < S4_T; r += 4) {
		const int64_t gi = (int64_t)bi * S4_T + r;
		if (gi >= nx || gy >= ny) continue;
		const double g = tile[tx][r];
		double vx = dxx[gi];
		if (vx == 0.0) vx = 1.0;                       // association.py:545-547
		const double expl = g * g * vx;                // gamma^2 dxx = variance explained by x_i
		const double vy = rs / ncells + expl;          // dyy (association.py:542)
		const double r2 = expl / vy;                   // dxy^2/(dxx dyy) (association.py:554)
		if (!isfinite(r2) || !isfinite(vy)) bad_nf = 1;
		if (r2 > 1.0 + 1e-8 || vy < 0.0) bad_rng = 1;
		const int64_t o = gi * ldo + gy;
		p_out[o] = (OutT)nrm_pvalue(r2, pl);
		stat_out[o] = (OutT)(return_dot ? g * vx : g);
		vary_out[o] = (OutT)vy;
	}
	if (flags) {
		if (bad_nf) atomicAdd(&flags[0], 1);
		if (bad_rng) atomicAdd(&flags[1], 1);
	}
}

static PvalPlan s4_to_dev(const nrm_pvalue_plan& p) {
	PvalPlan d;
	d.a = p.a;
	d.alpha = p.alpha;
	d.ln_front = p.ln_front;
	d.umax = p.umax;
	for (int j = 0; j < NRM_PCOEF; j++) d.coef[j] = p.coef[j];
	return d;
}

extern "C" int nrm_single4_sweep(const double* d_bt, const double* d_pt, int64_t ldb, const double* d_yy, const double* d_dxx,
								 int64_t nx, int64_t ny, int64_t m, int64_t n_cells, double dof, int return_dot, void* d_p,
								 void* d_stat, void* d_vary, int out_dtype, int64_t ldo, double* d_work, int32_t* d_flags, void* stream) {
	NRM_REQUIRE(nx > 0 && ny > 0 && m >= nx && n_cells > 0, "nrm_single4_sweep: bad sizes");
	NRM_REQUIRE(ldb >= m && ldo >= ny, "nrm_single4_sweep: pitch too small");
	NRM_REQUIRE(out_dtype == NRM_F32 || out_dtype == NRM_F64, "nrm_single4_sweep: bad out_dtype");
	NRM_REQUIRE(d_bt && d_pt && d_yy && d_dxx && d_p && d_stat && d_vary && d_work, "nrm_single4_sweep: null pointer");
	nrm_pvalue_plan plan;
	int rc = nrm_pvalue_plan_init(&plan, dof);
	if (rc) return rc;
	hipStream_t st = (hipStream_t)stream;
	double* rss = d_work;
	hipLaunchKernelGGL(k_s4_rss, dim3((unsigned)((ny + 3) / 4)), dim3(256), 0, st, d_bt, d_pt, ldb, d_yy, ny, m, rss);
	dim3 grid((unsigned)((ny + S4_T - 1) / S4_T), (unsigned)((nx + S4_T - 1) / S4_T));
	if (out_dtype == NRM_F64)
		hipLaunchKernelGGL(k_s4_sweep<double>, grid, dim3(256), 0, st, d_bt, ldb, rss, d_dxx, nx, ny, (double)n_cells, return_dot,
						   s4_to_dev(plan), (double*)d_p, (double*)d_stat, (double*)d_vary, ldo, d_flags);
	else
		hipLaunchKernelGGL(k_s4_sweep<float>, grid, dim3(256), 0, st, d_bt, ldb, rss, d_dxx, nx, ny, (double)n_cells, return_dot,
						   s4_to_dev(plan), (float*)d_p, (float*)d_stat, (float*)d_vary, ldo, d_flags);
	return nrm_check_launch("k_s4_sweep");
}
