// single=4 ("other groupings as covariates", association.py:421-576,926-980) in closed form: one multiple
// regression of every gene on A = [dx; dc] replaces the reference's per-grouping SVD loop.
// The heavy contractions A A^T, Y A^T and (Y A^T) N run on K2; this file is the per-pair sweep.
#include "nrm_pvalue.h"
#include "nrm_fix.h"

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

// The integer Gram engine's exact mean-product correction (nrm_fix.h) added to a raw dot matrix in place: dot[i, j] += sum_{s+t <= NS-2}
// u_i[s] u_j[t] / n from the row records of the rows (fr) and of the columns (fc).  K3 does this inside its sweeps; single=4 needs the
// corrected products themselves (they are the operand of the next contraction).
__global__ void __launch_bounds__(256) k_fix_dot(double* __restrict__ dot, int64_t ld, const double* __restrict__ fr, const double* __restrict__ fc,
												 int64_t rows, int64_t cols, int top, double inv_n) {
	const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
	if (j >= cols) return;
	const FixCol y = nrm_fix_col(fc + j * NRM_FIX_STRIDE);
#pragma unroll
	for (int it = 0; it < 4; it++) {  // 16 rows per workgroup, one of every four per wave
		const int64_t i = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 6) + 4 * it;
		if (i >= rows) break;
		double ux[5];
#pragma unroll
		for (int s = 0; s < 5; s++) ux[s] = fr[i * NRM_FIX_STRIDE + s];
		dot[i * ld + j] += nrm_fix_corr(ux, y, top, inv_n);
	}
}

extern "C" int nrm_gram_i8_fix_dot(double* d_dot, int64_t ldd, const double* d_fix_rows, const double* d_fix_cols, int64_t rows, int64_t cols,
								   int nslices, int64_t n_cells, void* stream) {
	NRM_REQUIRE(d_dot && d_fix_rows && d_fix_cols && rows > 0 && cols > 0 && ldd >= cols && n_cells > 0, "nrm_gram_i8_fix_dot: bad arguments");
	NRM_REQUIRE(nslices == 5 || nslices == 6, "nrm_gram_i8_fix_dot: 5 or 6 slices");
	dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 15) / 16));
	hipLaunchKernelGGL(k_fix_dot, grid, dim3(256), 0, (hipStream_t)stream, d_dot, ldd, d_fix_rows, d_fix_cols, rows, cols, nslices - 2, 1.0 / (double)n_cells);
	return nrm_check_launch("k_fix_dot");
}

// What the single=4 sweep needs to certify P-values whose products Y~ X~^T came from the integer engine (fy == nullptr: fp64 Gram
// kernel, nothing to certify).  |delta (y~ . x~_j)| <= e_y |y~| |x~_j| with e_y = K c_y c* + g_y + g* (c*, g*: the largest c and g
// among the design rows, nrm_fix.h), so the regression coefficient B_yi = sum_j G_yj N_ji moves by at most e_y |y~| sum_j |x~_j| |N_ji|
// and the partial correlation by  dr <= 2 e_y kappa_i |y~| / sqrt(n vary_iy),  kappa_i = sum_j |x~_j| |N_ji| / sqrt(N_ii)  (the factor 2
// covers the matching change of the residual sum of squares).  A pair is counted when that could move its P-value by more than the
// budget, with the P-value's sensitivity bounded as in nrm_fix_bound.
struct S4Guard {
	const double* fy;     // (ny, NRM_FIX_STRIDE) row records of the genes, or nullptr
	const double* kappa;  // (nx)
	const double* yy;     // (ny) |y~|^2
	int32_t* gene_hits;   // (ny) or nullptr: set to 1 for every gene with a counted pair (the caller redoes only those genes in fp64)
	double kconst, cstar, gstar, budget, dof;
};

#define S4_T 64
template <typename OutT>
__global__ void __launch_bounds__(256) k_s4_sweep(const double* __restrict__ bt, int64_t ldb, const double* __restrict__ rss,
												  const double* __restrict__ dxx, int64_t nx, int64_t ny, double ncells,
												  int return_dot, PvalPlan pl, OutT* __restrict__ p_out, OutT* __restrict__ stat_out,
												  OutT* __restrict__ vary_out, int64_t ldo, int32_t* __restrict__ flags, S4Guard gd) {
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
	int bad_nf = 0, bad_rng = 0, bad_fix = 0;
	float worst = 0.f;
	double ey = 0.0, sqrt_dof = 0.0;
	if (gd.fy && gy < ny) {
		ey = 2.0 * (fma(gd.kconst * gd.fy[gy * NRM_FIX_STRIDE + 5], gd.cstar, gd.fy[gy * NRM_FIX_STRIDE + 6] + gd.gstar)) * sqrt(gd.yy[gy] / ncells);
		sqrt_dof = sqrt(gd.dof);
	}
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
		const OutT pv = (OutT)nrm_pvalue(r2, pl);
		p_out[o] = pv;
		if (gd.fy) {  // the integer engine's products: could their error move this P-value by more than the budget?
			const double dr = ey * gd.kappa[gi] * rsqrt(vy);
			const double ar = (double)(sqrtf((float)r2) * 1.0000002f);
			const double om = fmax(1.0 - r2, 1e-150);
			const double num = dr * fma(gd.dof, ar, sqrt_dof), den = om * om;
			if (num > gd.budget * den || !(num == num)) {
				const double lo = fmax(ar - dr, 0.0);
				if (pv != (OutT)0 || nrm_pvalue(lo * lo, pl) != 0.0) {  // (a P-value that is 0 on the whole interval is exempt)
					bad_fix++;
					if (gd.gene_hits) gd.gene_hits[gy] = 1;
				}
			} else
				worst = fmaxf(worst, __fdividef((float)num, (float)den));
		}
		stat_out[o] = (OutT)(return_dot ? g * vx : g);
		vary_out[o] = (OutT)vy;
	}
	if (flags) {
		if (bad_nf) atomicAdd(&flags[0], 1);
		if (bad_rng) atomicAdd(&flags[1], 1);
		if (gd.fy) {
			if (bad_fix && __hip_atomic_load(flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (1 << 30)) atomicAdd(&flags[2], bad_fix);
			const int w = __float_as_int(worst);  // non-negative floats order like their bit patterns
			if (w > __hip_atomic_load(flags + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&flags[3], w);
		}
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

static int single4_sweep_impl(const double* d_bt, const double* d_pt, int64_t ldb, const double* d_yy, const double* d_dxx,
							  int64_t nx, int64_t ny, int64_t m, int64_t n_cells, double dof, int return_dot, void* d_p,
							  void* d_stat, void* d_vary, int out_dtype, int64_t ldo, double* d_work, int32_t* d_flags, S4Guard gd, void* stream) {
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
						   s4_to_dev(plan), (double*)d_p, (double*)d_stat, (double*)d_vary, ldo, d_flags, gd);
	else
		hipLaunchKernelGGL(k_s4_sweep<float>, grid, dim3(256), 0, st, d_bt, ldb, rss, d_dxx, nx, ny, (double)n_cells, return_dot,
						   s4_to_dev(plan), (float*)d_p, (float*)d_stat, (float*)d_vary, ldo, d_flags, gd);
	return nrm_check_launch("k_s4_sweep");
}

extern "C" int nrm_single4_sweep(const double* d_bt, const double* d_pt, int64_t ldb, const double* d_yy, const double* d_dxx,
								 int64_t nx, int64_t ny, int64_t m, int64_t n_cells, double dof, int return_dot, void* d_p,
								 void* d_stat, void* d_vary, int out_dtype, int64_t ldo, double* d_work, int32_t* d_flags, void* stream) {
	const S4Guard none = {nullptr, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, 0.0, dof};
	return single4_sweep_impl(d_bt, d_pt, ldb, d_yy, d_dxx, nx, ny, m, n_cells, dof, return_dot, d_p, d_stat, d_vary, out_dtype, ldo, d_work, d_flags, none, stream);
}

// The same for products that came from the integer Gram engine: d_fix_y = the genes' row records, d_kappa (nx) and c*, g* as described
// at S4Guard; d_flags then has 4 entries (non-finite, R^2 out of range, pairs not certified, largest error estimate as float bits);
// d_gene_hits (ny int32, zeroed by the caller) or NULL: set to 1 for every gene with a pair that was counted.
extern "C" int nrm_single4_sweep_guarded(const double* d_bt, const double* d_pt, int64_t ldb, const double* d_yy, const double* d_dxx,
										 int64_t nx, int64_t ny, int64_t m, int64_t n_cells, double dof, int return_dot, void* d_p,
										 void* d_stat, void* d_vary, int out_dtype, int64_t ldo, double* d_work, int32_t* d_flags,
										 const double* d_fix_y, const double* d_kappa, double cstar, double gstar, int nslices, double budget,
										 int32_t* d_gene_hits, void* stream) {
	NRM_REQUIRE(d_fix_y && d_kappa && d_flags && (nslices == 5 || nslices == 6) && budget > 0.0, "nrm_single4_sweep_guarded: bad guard arguments");
	double k = 0.0, w256 = 1.0;
	for (int w = 0; w <= nslices - 2; w++, w256 *= 256.0) k += (w + 1) * w256;
	const S4Guard gd = {d_fix_y, d_kappa, d_yy, d_gene_hits, k, cstar, gstar, budget, dof};
	return single4_sweep_impl(d_bt, d_pt, ldb, d_yy, d_dxx, nx, ny, m, n_cells, dof, return_dot, d_p, d_stat, d_vary, out_dtype, ldo, d_work, d_flags, gd, stream);
}

// ---- the small steps around the Newton-Schulz inverse of M~ (normalisr_amd/single4.py: _spd_inverse_device) ------------------------------------
// X <- X (2 I - M X) runs its two 1024^3 products per step on the fp64 Gram kernel; what sits between them -- symmetrising the Gram
// kernel's upper tiles, the start matrix, ||I - M X||_F, a transpose, 2 X - X T, the inverse's diagonal / kappa / 1-norm -- was a
// dozen torch element-wise and reduction kernels per step (rounds 3-4: 144 + 120 + 48 ... launches in a configs[3] call).  Plain
// tile kernels; every reduction in a fixed order (rows by one workgroup each, then one workgroup over the rows).
namespace {

// mp (nxp, nxp) = the symmetric matrix whose upper triangle (i <= j < nx) is in m (tiles on / above the diagonal of a symmetric K2
// launch), zero outside nx x nx; tiles of 32 x 32, the mirror written from the transpose in LDS
__global__ void __launch_bounds__(256) k_spd_sym(const double* __restrict__ m, int64_t ldm, int64_t nx, int64_t nxp, double* __restrict__ mp) {
	__shared__ double tile[32][33];
	const int bi = blockIdx.y, bj = blockIdx.x;
	if (bj < bi) return;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
	for (int r = ty; r < 32; r += 8) {
		const int64_t i = (int64_t)bi * 32 + r, j = (int64_t)bj * 32 + tx;
		double v = 0.0;
		if (i < nx && j < nx) v = i <= j ? m[i * ldm + j] : m[j * ldm + i];  // (inside a diagonal tile the lower half comes from the upper)
		tile[r][tx] = v;
		mp[i * nxp + j] = v;
	}
	__syncthreads();
	if (bj > bi) {
#pragma unroll
		for (int r = ty; r < 32; r += 8) mp[((int64_t)bj * 32 + r) * nxp + (int64_t)bi * 32 + tx] = tile[tx][r];
	}
}

// per row of a symmetric matrix: out0[i] = sum_j |a_ij| (= the column's: the 1-norm's candidates), out1[i] = sum_j |a_ij| wgt_j (or not), out2[i] = a_ii
__global__ void __launch_bounds__(256) k_spd_rows(const double* __restrict__ a, int64_t ld, int64_t nx, const double* __restrict__ ss, double* __restrict__ out0,
												  double* __restrict__ out1, double* __restrict__ out2) {
	__shared__ double red[2][4];
	const int64_t i = blockIdx.x;
	double s0 = 0.0, s1 = 0.0;
	for (int64_t j = threadIdx.x; j < nx; j += 256) {
		const double v = fabs(a[i * ld + j]);
		s0 += v;
		if (ss) s1 = fma(v, sqrt(ss[j]), s1);
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		s0 += __shfl_down(s0, o, 64);
		s1 += __shfl_down(s1, o, 64);
	}
	if ((threadIdx.x & 63) == 0) {
		red[0][threadIdx.x >> 6] = s0;
		red[1][threadIdx.x >> 6] = s1;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		out0[i] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
		if (out1) out1[i] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
		if (out2) out2[i] = a[i * ld + i];
	}
}

// one workgroup: scal[0] = max_i rowsum[i] (||M||_1), scal[1] = mean of the diagonal; the padding block of mp (rows nx .. nxp) gets that
// mean on its diagonal (a multiple of the identity inside the spectrum's range: it does not slow the iteration down)
__global__ void __launch_bounds__(1024) k_spd_scale(const double* __restrict__ rowsum, const double* __restrict__ diag, int64_t nx, int64_t nxp, double* __restrict__ mp,
													 double* __restrict__ scal) {
	__shared__ double mx[1024], sm[1024];
	const int t = threadIdx.x;
	double a = 0.0, b = 0.0;
	bool bad = false;
	for (int64_t i = t; i < nx; i += 1024) {
		bad |= !(rowsum[i] == rowsum[i]);
		a = fmax(a, rowsum[i]);
		b += diag[i];
	}
	mx[t] = bad ? NAN : a;
	sm[t] = b;
	__syncthreads();
	for (int o = 512; o > 0; o >>= 1) {
		if (t < o) {
			mx[t] = (mx[t] != mx[t] || mx[t + o] != mx[t + o]) ? NAN : fmax(mx[t], mx[t + o]);
			sm[t] += sm[t + o];
		}
		__syncthreads();
	}
	const double mean = sm[0] / (double)nx;
	if (t == 0) {
		scal[0] = mx[0];
		scal[1] = mean;
	}
	for (int64_t i = nx + t; i < nxp; i += 1024) mp[i * nxp + i] = mean;
}

// the start of the iteration: x = diag(1 / mp_ii) (diagonal != 0) or I / scal[0]
__global__ void __launch_bounds__(256) k_spd_start(const double* __restrict__ mp, int64_t nxp, int diagonal, const double* __restrict__ scal, double* __restrict__ x) {
	const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= nxp * nxp) return;
	const int64_t i = e / nxp, j = e - i * nxp;
	x[e] = i == j ? (diagonal ? 1.0 / mp[i * nxp + i] : 1.0 / scal[0]) : 0.0;
}

// tt = t^T (32 x 32 tiles through LDS) and, on the way, the squares of I - t row block by row block: part[tile] = sum over the tile
__global__ void __launch_bounds__(256) k_spd_transpose_res(const double* __restrict__ t, int64_t nxp, double* __restrict__ tt, double* __restrict__ part) {
	__shared__ double tile[32][33];
	__shared__ double red[4];
	const int bi = blockIdx.y, bj = blockIdx.x;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	double s = 0.0;
#pragma unroll
	for (int r = ty; r < 32; r += 8) {
		const int64_t i = (int64_t)bi * 32 + r, j = (int64_t)bj * 32 + tx;
		const double v = t[i * nxp + j];
		tile[r][tx] = v;
		const double d = (i == j ? 1.0 : 0.0) - v;
		s = fma(d, d, s);
	}
	__syncthreads();
#pragma unroll
	for (int r = ty; r < 32; r += 8) tt[((int64_t)bj * 32 + r) * nxp + (int64_t)bi * 32 + tx] = tile[tx][r];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) part[(int64_t)bi * gridDim.x + bj] = ((red[0] + red[1]) + red[2]) + red[3];
}

// one workgroup: res[0] = sqrt(sum of part[0 .. count)) in a fixed order
__global__ void __launch_bounds__(1024) k_spd_res(const double* __restrict__ part, int64_t count, double* __restrict__ res) {
	__shared__ double sm[1024];
	const int t = threadIdx.x;
	double b = 0.0;
	for (int64_t i = t; i < count; i += 1024) b += part[i];
	sm[t] = b;
	__syncthreads();
	for (int o = 512; o > 0; o >>= 1) {
		if (t < o) sm[t] += sm[t + o];
		__syncthreads();
	}
	if (t == 0) res[0] = sqrt(sm[0]);
}

// x = 2 x - xt
__global__ void __launch_bounds__(256) k_spd_update(double* __restrict__ x, const double* __restrict__ xt, int64_t count) {
	const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
	if (e + 1 < count) {
		double2 a = *reinterpret_cast<const double2*>(x + e);
		const double2 b = *reinterpret_cast<const double2*>(xt + e);
		a.x = 2.0 * a.x - b.x;
		a.y = 2.0 * a.y - b.y;
		*reinterpret_cast<double2*>(x + e) = a;
	} else if (e < count)
		x[e] = 2.0 * x[e] - xt[e];
}

// out = (x + x^T) / 2 inside nx x nx, zero outside (out != x: the tiles of a pair are read by two workgroups)
__global__ void __launch_bounds__(256) k_spd_finish(const double* __restrict__ x, int64_t nx, int64_t nxp, double* __restrict__ out) {
	__shared__ double tile[32][33];
	const int bi = blockIdx.y, bj = blockIdx.x;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
	for (int r = ty; r < 32; r += 8) tile[r][tx] = x[((int64_t)bj * 32 + r) * nxp + (int64_t)bi * 32 + tx];  // the mirror tile
	__syncthreads();
#pragma unroll
	for (int r = ty; r < 32; r += 8) {
		const int64_t i = (int64_t)bi * 32 + r, j = (int64_t)bj * 32 + tx;
		out[i * nxp + j] = (i < nx && j < nx) ? 0.5 * (x[i * nxp + j] + tile[tx][r]) : 0.0;
	}
}

}  // namespace

// d_m (.., ldm): a symmetric positive definite nx x nx matrix as a symmetric K2 launch leaves it (entries i <= j valid).  d_mp (nxp, nxp), nxp a
// multiple of 128 >= nx: the full symmetric matrix, padded with mean(diag) on the diagonal; d_scal[0] = ||M||_1, d_scal[1] = mean(diag) (NaN when
// the matrix holds one); d_work: 2 nxp doubles.
extern "C" int nrm_spd_prepare(const double* d_m, int64_t ldm, int64_t nx, int64_t nxp, double* d_mp, double* d_scal, double* d_work, void* stream) {
	NRM_REQUIRE(d_m && d_mp && d_scal && d_work && nx > 0 && nxp >= nx && nxp % 32 == 0 && ldm >= nx, "nrm_spd_prepare: bad arguments");
	hipStream_t st = (hipStream_t)stream;
	const unsigned nt = (unsigned)(nxp / 32);
	hipLaunchKernelGGL(k_spd_sym, dim3(nt, nt), dim3(256), 0, st, d_m, ldm, nx, nxp, d_mp);
	hipLaunchKernelGGL(k_spd_rows, dim3((unsigned)nx), dim3(256), 0, st, d_mp, nxp, nx, nullptr, d_work, nullptr, d_work + nxp);
	hipLaunchKernelGGL(k_spd_scale, dim3(1), dim3(1024), 0, st, d_work, d_work + nxp, nx, nxp, d_mp, d_scal);
	return nrm_check_launch("nrm_spd_prepare");
}

// d_x (nxp, nxp) = the start of the Newton-Schulz iteration: diag(1 / M_ii) (diagonal != 0) or I / ||M||_1
extern "C" int nrm_spd_start(const double* d_mp, int64_t nxp, int diagonal, const double* d_scal, double* d_x, void* stream) {
	NRM_REQUIRE(d_mp && d_scal && d_x && nxp > 0, "nrm_spd_start: bad arguments");
	hipLaunchKernelGGL(k_spd_start, dim3((unsigned)((nxp * nxp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_mp, nxp, diagonal, d_scal, d_x);
	return nrm_check_launch("k_spd_start");
}

// d_tt = d_t^T and d_res[0] = ||I - d_t||_F for a (nxp, nxp) matrix; d_work: (nxp / 32)^2 doubles
extern "C" int nrm_spd_transpose_residual(const double* d_t, int64_t nxp, double* d_tt, double* d_res, double* d_work, void* stream) {
	NRM_REQUIRE(d_t && d_tt && d_res && d_work && nxp > 0 && nxp % 32 == 0, "nrm_spd_transpose_residual: bad arguments");
	hipStream_t st = (hipStream_t)stream;
	const unsigned nt = (unsigned)(nxp / 32);
	hipLaunchKernelGGL(k_spd_transpose_res, dim3(nt, nt), dim3(256), 0, st, d_t, nxp, d_tt, d_work);
	hipLaunchKernelGGL(k_spd_res, dim3(1), dim3(1024), 0, st, d_work, (int64_t)nt * nt, d_res);
	return nrm_check_launch("nrm_spd_transpose_residual");
}

// d_x = 2 d_x - d_xt (count doubles, 16-byte aligned)
extern "C" int nrm_spd_update(double* d_x, const double* d_xt, int64_t count, void* stream) {
	NRM_REQUIRE(d_x && d_xt && count > 0, "nrm_spd_update: bad arguments");
	hipLaunchKernelGGL(k_spd_update, dim3((unsigned)((count / 2 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, d_xt, count);
	return nrm_check_launch("k_spd_update");
}

// d_n (nxp, nxp) = (d_x + d_x^T) / 2 inside nx x nx, zero in the padding; d_small (3, nx): the diagonal of d_n, kappa's numerator
// sum_j |N_ij| sqrt(d_ss[j]) (S4Guard above) and the absolute row sums (their maximum is ||N||_1)
extern "C" int nrm_spd_finish(const double* d_x, int64_t nx, int64_t nxp, const double* d_ss, double* d_n, double* d_small, void* stream) {
	NRM_REQUIRE(d_x && d_n && d_small && d_ss && d_n != d_x && nx > 0 && nxp >= nx && nxp % 32 == 0, "nrm_spd_finish: bad arguments");
	hipStream_t st = (hipStream_t)stream;
	const unsigned nt = (unsigned)(nxp / 32);
	hipLaunchKernelGGL(k_spd_finish, dim3(nt, nt), dim3(256), 0, st, d_x, nx, nxp, d_n);
	hipLaunchKernelGGL(k_spd_rows, dim3((unsigned)nx), dim3(256), 0, st, d_n, nxp, nx, d_ss, d_small + 2 * nx, d_small + nx, d_small);
	return nrm_check_launch("nrm_spd_finish");
}

// ---- what the sweep needs of N~ = M~^-1, taken on the device (a resident single=4 step has nothing on the host: single4.Single4Plan) --------------
// d_small (3, nx) of nrm_spd_finish: row 0 = the diagonal d_i of N~.  d_dxx[i] = 1 / (n d_i) = the partial variance of design row i given all other
// rows and the covariates (association.py:539-540 in the closed form of single4.py); d_varx[i] the same with the reference's 0 -> 1 (:546-547);
// flags[5] += design rows whose d_i is not a positive finite number (the rows are linearly dependent given the covariates: the host raises LinAlgError).
__global__ void __launch_bounds__(256) k_s4_design_scalars(const double* __restrict__ small, int64_t nx, double n_cells, double* __restrict__ dxx,
															double* __restrict__ varx, int32_t* __restrict__ flags) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= nx) return;
	const double d = small[i], ka = small[nx + i], rs = small[2 * nx + i];
	if (!(isfinite(d) && d > 0.0 && isfinite(ka) && isfinite(rs))) atomicAdd(&flags[5], 1);
	const double v = 1.0 / (n_cells * d);
	dxx[i] = v;
	varx[i] = v == 0.0 ? 1.0 : v;
}

extern "C" int nrm_single4_design_scalars(const double* d_small, int64_t nx, int64_t n_cells, double* d_dxx, double* d_varx, int32_t* d_flags, void* stream) {
	NRM_REQUIRE(d_small && d_dxx && d_varx && d_flags && nx > 0 && n_cells > 0, "nrm_single4_design_scalars: bad arguments");
	hipLaunchKernelGGL(k_s4_design_scalars, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_small, nx, (double)n_cells, d_dxx, d_varx, d_flags);
	return nrm_check_launch("k_s4_design_scalars");
}
