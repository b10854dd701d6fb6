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
