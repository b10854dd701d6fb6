// normvar (reference norm.py:131-289): variance normalisation with PER-GENE weighted covariates -- the step right
// before de/coex in the Normalisr pipeline.  Gene g is scaled by e_gk = w_k^wt_g and the covariates C * e_g are
// removed from it: a different (nc x nc) OLS per gene.  The reference loops over genes; here
//     M_g = sum_k e_gk^2 C_k C_k^T   and   a_g = sum_k e_gk^2 y_gk C_k
// are rows of two Gram contractions on the fp64 matrix cores (K2): U P^T and V C^T with U = e^2, V = e^2 y and
// P = the nc(nc+1)/2 products C_c * C_c'.  This file holds the two HBM-bound element-wise passes around them.
#include "nrm_common.h"
#include "nrm_jacobi.h"

#include "nrm_exp2_tab.h"

#define NV_R 4

static bool nv_aligned(const void* d_y, int y_dtype, int64_t ldy, const double* d_lnw, const double* d_c, int64_t ldc);

// e = exp(x) for the per-gene cell weights w_k^wt_g = exp(wt_g ln w_k) (norm.py:245).  Both passes over the matrix are bound by the fp64 vector ALU, and
// the library's exp() was most of it (round-5 verdict: normvar at 0.14 of the HBM roofline it is priced against).  Here: x = (64 k + j) ln2/64 + r with
// |r| <= ln2/128, exp(x) = 2^k 2^(j/64) exp(r); 2^(j/64) from a 64-entry table (correctly rounded, csrc/nrm_exp2_tab.h; the workgroup's copy in LDS),
// exp(r) as its Taylor polynomial of degree 5 (the term left out, r^6/720, is below 4e-17), ln2/64 in two parts so that the reduction is exact for
// |k 64 + j| < 2^21.  Fourteen instructions, eleven of them at the fp64 rate, no branch; relative error <= 3e-16 on |x| <= 700 (held to
// numpy's exp through nrm_normvar_exp_probe: tests/test_gpu_round6.py), overflow / underflow as ldexp gives them.
__device__ __forceinline__ double nv_exp(double x, const double* __restrict__ tab) {
	const double kd = rint(x * 92.33248261689366);  // 64 / ln 2
	const int ki = (int)kd;
	double r = fma(kd, -0x1.62e42fee00000p-7, x);  // ln2/64, high part (32 bits: kd * hi is exact)
	r = fma(kd, -0x1.a39ef35793c76p-39, r);
	double p = fma(r, 1.0 / 120, 1.0 / 24);
	p = fma(p, r, 1.0 / 6);
	p = fma(p, r, 0.5);
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	return ldexp(tab[ki & 63] * p, ki >> 6);
}

// Four consecutive elements of a row as doubles.  ALIGNED (the launcher: 16-byte aligned rows): one 16-byte load (two for doubles); otherwise element loads --
// a template switch, not a branch.  Either way the four are independent loads: round 6 found both passes of normvar bound by the LATENCY of one dependent
// 4-byte load per thread and iteration (k_nv_apply waited for HBM once per row and cell: 39 x 4 round trips of ~1.5 us per workgroup = the 0.27 ms it took),
// not by exp() and the multiply-adds as rounds 4-5 had it; four cells per thread and iteration, every load of an iteration issued before the first use.
template <typename T, bool ALIGNED>
__device__ __forceinline__ void nv_ld4(const T* __restrict__ p, double (&v)[4]) {
	if constexpr (ALIGNED && sizeof(T) == 4) {
		const float4 t = *reinterpret_cast<const float4*>(p);
		v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
	} else if constexpr (ALIGNED) {
		const double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2);
		v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
	} else {
#pragma unroll
		for (int j = 0; j < 4; j++) v[j] = (double)p[j];
	}
}

__device__ __forceinline__ void nv_exp_table(double* tab, int tid) {  // (before the workgroup's first barrier)
	if (tid < 64) tab[tid] = kNrmExp2_64[tid];
}

__global__ void __launch_bounds__(256) k_nv_exp_probe(const double* __restrict__ x, int64_t count, double* __restrict__ out) {
	__shared__ double tab[64];
	nv_exp_table(tab, threadIdx.x);
	__syncthreads();
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) out[i] = nv_exp(x[i], tab);
}

// d_out[i] = the kernels' exp(d_x[i]) (a probe for the tests: the function has no other way out of the library)
extern "C" int nrm_normvar_exp_probe(const double* d_x, int64_t count, double* d_out, void* stream) {
	NRM_REQUIRE(d_x && d_out && count > 0, "nrm_normvar_exp_probe: bad arguments");
	hipLaunchKernelGGL(k_nv_exp_probe, dim3(256), dim3(256), 0, (hipStream_t)stream, d_x, count, d_out);
	return nrm_check_launch("k_nv_exp_probe");
}

__device__ __forceinline__ double nv_wave_sum(double v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;
}

// pass 1: U = e^2, V = e^2 y (fp64, zero padded), s1 = sum y e, s2 = sum (y e)^2      (norm.py:244-249)
template <typename T>
__global__ void __launch_bounds__(256) k_nv_weights(const T* __restrict__ y, int64_t rows, int64_t n, int64_t ldy,
													const double* __restrict__ lnw, const double* __restrict__ wt, double* __restrict__ U,
													double* __restrict__ V, int64_t ldo, double* __restrict__ s1, double* __restrict__ s2) {
	__shared__ double sm[4][2 * NV_R];
	__shared__ double tab[64];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	nv_exp_table(tab, tid);
	__syncthreads();
	const int64_t row0 = (int64_t)blockIdx.x * NV_R;
	double a1[NV_R], a2[NV_R], ex[NV_R];
	bool live[NV_R];
#pragma unroll
	for (int r = 0; r < NV_R; r++) {
		live[r] = row0 + r < rows;
		ex[r] = live[r] ? wt[row0 + r] : 0.0;
		a1[r] = a2[r] = 0.0;
	}
	for (int64_t k = tid; k < ldo; k += 256) {
		const double lw = k < n ? lnw[k] : 0.0;
#pragma unroll
		for (int r = 0; r < NV_R; r++) {
			double u = 0.0, v = 0.0;
			if (live[r] && k < n) {
				const double e = ex[r] == 0.0 ? 1.0 : nv_exp(ex[r] * lw, tab);  // w**wt, exactly 1 for wt == 0 (norm.py:245)
				const double yv = (double)y[(row0 + r) * ldy + k];
				const double yp = yv * e;
				u = e * e;
				v = u * yv;
				a1[r] += yp;
				a2[r] = fma(yp, yp, a2[r]);
			}
			U[(row0 + r) * ldo + k] = u;
			V[(row0 + r) * ldo + k] = v;
		}
	}
#pragma unroll
	for (int r = 0; r < NV_R; r++) {
		const double x1 = nv_wave_sum(a1[r]), x2 = nv_wave_sum(a2[r]);
		if (lane == 0) {
			sm[wid][2 * r] = x1;
			sm[wid][2 * r + 1] = x2;
		}
	}
	__syncthreads();
	if (tid < NV_R) {
		s1[row0 + tid] = sm[0][2 * tid] + sm[1][2 * tid] + sm[2][2 * tid] + sm[3][2 * tid];
		s2[row0 + tid] = sm[0][2 * tid + 1] + sm[1][2 * tid + 1] + sm[2][2 * tid + 1] + sm[3][2 * tid + 1];
	}
}

// pass 2: out = scale_g * e_gk * (y_gk - sum_c b_gc C_ck)      (norm.py:157 per gene, :259)
// Four cells per thread and iteration, the loads of all NV_R rows, the weights and (batch by batch) the covariates issued before anything is used: the first
// form loaded one 4-byte value per row and cell and waited for each (see nv_ld4).
template <typename T, typename OutT, bool ALIGNED>
__global__ void __launch_bounds__(256) k_nv_apply(const T* __restrict__ y, int64_t rows, int64_t n, int64_t ldy, const double* __restrict__ lnw,
												  const double* __restrict__ wt, const double* __restrict__ c, int nc, int64_t ldc,
												  const double* __restrict__ b, const double* __restrict__ scale, OutT* __restrict__ out,
												  int64_t ldo, int32_t* __restrict__ flags) {
	__shared__ double s_b[NV_R][64];
	__shared__ double tab[64];
	bool bad = false;
	const int tid = threadIdx.x;
	nv_exp_table(tab, tid);
	const int64_t row0 = (int64_t)blockIdx.x * NV_R;
	for (int i = tid; i < NV_R * nc; i += 256) {
		const int r = i / nc, q = i % nc;
		s_b[r][q] = row0 + r < rows ? b[(row0 + r) * nc + q] : 0.0;
	}
	__syncthreads();
	double ex[NV_R], sc[NV_R];
	const T* yr[NV_R];
#pragma unroll
	for (int r = 0; r < NV_R; r++) {
		const bool live = row0 + r < rows;
		ex[r] = live ? wt[row0 + r] : 0.0;
		sc[r] = live ? scale[row0 + r] : 0.0;
		yr[r] = y + (live ? row0 + r : rows - 1) * ldy;  // (rows past the end read the last row: every load is issued whatever the row, nothing of theirs is stored)
	}
	auto finish = [&](int r, double lw, double yv, double fit) {
		const double e = ex[r] == 0.0 ? 1.0 : nv_exp(ex[r] * lw, tab);
		const OutT o = (OutT)(sc[r] * e * (yv - fit));
		bad |= !(fabs((double)o) <= 1.7976931348623157e308);
		return o;
	};
	const int64_t n4 = n & ~(int64_t)3;
	for (int64_t k = (int64_t)tid * 4; k < n4; k += 1024) {
		double lw[4], yv[NV_R][4], fit[NV_R][4];
		nv_ld4<double, ALIGNED>(lnw + k, lw);
#pragma unroll
		for (int r = 0; r < NV_R; r++) {
			nv_ld4<T, ALIGNED>(yr[r] + k, yv[r]);
#pragma unroll
			for (int v = 0; v < 4; v++) fit[r][v] = 0.0;
		}
		for (int q = 0; q < nc; q++) {
			double cv[4];
			nv_ld4<double, ALIGNED>(c + (int64_t)q * ldc + k, cv);
#pragma unroll
			for (int r = 0; r < NV_R; r++)
#pragma unroll
				for (int v = 0; v < 4; v++) fit[r][v] = fma(s_b[r][q], cv[v], fit[r][v]);
		}
#pragma unroll
		for (int r = 0; r < NV_R; r++) {
			if (row0 + r >= rows) continue;
			OutT o[4];
#pragma unroll
			for (int v = 0; v < 4; v++) o[v] = finish(r, lw[v], yv[r][v], fit[r][v]);
			OutT* dst = out + (row0 + r) * ldo + k;
			if constexpr (ALIGNED) {
				typedef OutT ov_t __attribute__((ext_vector_type(16 / sizeof(OutT))));
#pragma unroll
				for (int h = 0; h < 4; h += 16 / (int)sizeof(OutT)) {
					ov_t t;
#pragma unroll
					for (int j = 0; j < 16 / (int)sizeof(OutT); j++) t[j] = o[h + j];
					*reinterpret_cast<ov_t*>(dst + h) = t;
				}
			} else {
#pragma unroll
				for (int v = 0; v < 4; v++) dst[v] = o[v];
			}
		}
	}
	for (int64_t k = n4 + tid; k < n; k += 256) {  // the last n % 4 cells
		const double lw = lnw[k];
		double fit[NV_R];
#pragma unroll
		for (int r = 0; r < NV_R; r++) fit[r] = 0.0;
		for (int q = 0; q < nc; q++) {
			const double cv = c[(int64_t)q * ldc + k];
#pragma unroll
			for (int r = 0; r < NV_R; r++) fit[r] = fma(s_b[r][q], cv, fit[r]);
		}
#pragma unroll
		for (int r = 0; r < NV_R; r++)
			if (row0 + r < rows) out[(row0 + r) * ldo + k] = finish(r, lw, (double)yr[r][k], fit[r]);
	}
	// the reference asserts that its result is finite (norm.py:286): counted here, where the values are at hand (flags[1] += waves with one that is not)
	if (flags && __ballot(bad) && (tid & 63) == 0) atomicAdd(&flags[1], 1);
}

// normvar1 with explicit per-gene cell weights (norm.py:150-153: row g is residualised against dc * w2[g]):
//     out_gk = y_gk - w2_gk sum_c b_gc C_ck
template <typename T, typename OutT>
__global__ void __launch_bounds__(256) k_nv_apply_w2(const T* __restrict__ y, int64_t rows, int64_t n, int64_t ldy, const double* __restrict__ w2,
													 int64_t ldw, const double* __restrict__ c, int nc, int64_t ldc, const double* __restrict__ b,
													 OutT* __restrict__ out, int64_t ldo) {
	__shared__ double s_b[NV_R][64];
	const int tid = threadIdx.x;
	const int64_t row0 = (int64_t)blockIdx.x * NV_R;
	for (int i = tid; i < NV_R * nc; i += 256) {
		const int r = i / nc, q = i % nc;
		s_b[r][q] = row0 + r < rows ? b[(row0 + r) * nc + q] : 0.0;
	}
	__syncthreads();
	for (int64_t k = tid; k < n; k += 256) {
		double fit[NV_R];
#pragma unroll
		for (int r = 0; r < NV_R; r++) fit[r] = 0.0;
		for (int q = 0; q < nc; q++) {
			const double cv = c[(int64_t)q * ldc + k];
#pragma unroll
			for (int r = 0; r < NV_R; r++) fit[r] = fma(s_b[r][q], cv, fit[r]);
		}
#pragma unroll
		for (int r = 0; r < NV_R; r++)
			if (row0 + r < rows) out[(row0 + r) * ldo + k] = (OutT)((double)y[(row0 + r) * ldy + k] - w2[(row0 + r) * ldw + k] * fit[r]);
	}
}

extern "C" int nrm_normvar_apply_w2(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, const double* d_w2, int64_t ldw,
									const double* d_c, int64_t nc, int64_t ldc, const double* d_b, void* d_out, int out_dtype, int64_t ldo, void* stream) {
	NRM_REQUIRE((y_dtype == NRM_F32 || y_dtype == NRM_F64) && (out_dtype == NRM_F32 || out_dtype == NRM_F64), "nrm_normvar_apply_w2: bad dtype");
	NRM_REQUIRE(rows > 0 && n > 0 && ldy >= n && ldw >= n && ldo >= n && nc > 0 && nc <= 64 && ldc >= n, "Unmatched gene or cell counts.");
	NRM_REQUIRE(d_y && d_w2 && d_c && d_b && d_out, "nrm_normvar_apply_w2: null pointer");
	dim3 grid((unsigned)((rows + NV_R - 1) / NV_R));
	hipStream_t st = (hipStream_t)stream;
#define NV_LAUNCH(TY, TO) hipLaunchKernelGGL((k_nv_apply_w2<TY, TO>), grid, dim3(256), 0, st, (const TY*)d_y, rows, n, ldy, d_w2, ldw, d_c, (int)nc, ldc, d_b, (TO*)d_out, ldo)
	if (y_dtype == NRM_F64 && out_dtype == NRM_F64) NV_LAUNCH(double, double);
	else if (y_dtype == NRM_F64) NV_LAUNCH(double, float);
	else if (out_dtype == NRM_F64) NV_LAUNCH(float, double);
	else NV_LAUNCH(float, float);
#undef NV_LAUNCH
	return nrm_check_launch("k_nv_apply_w2");
}

extern "C" int nrm_normvar_weights(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, const double* d_lnw, const double* d_wt,
								   double* d_u, double* d_v, int64_t ldo, int64_t rows_pad, double* d_s1, double* d_s2, void* stream) {
	NRM_REQUIRE(y_dtype == NRM_F32 || y_dtype == NRM_F64, "nrm_normvar_weights: bad dtype");
	NRM_REQUIRE(rows > 0 && n > 0 && ldy >= n && ldo >= n && rows_pad >= rows && rows_pad % NV_R == 0, "Unmatched gene or cell counts.");
	NRM_REQUIRE(d_y && d_lnw && d_wt && d_u && d_v && d_s1 && d_s2, "nrm_normvar_weights: null pointer");
	dim3 grid((unsigned)(rows_pad / NV_R));
	if (y_dtype == NRM_F64)
		hipLaunchKernelGGL(k_nv_weights<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)d_y, rows, n, ldy, d_lnw, d_wt, d_u, d_v, ldo, d_s1, d_s2);
	else
		hipLaunchKernelGGL(k_nv_weights<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)d_y, rows, n, ldy, d_lnw, d_wt, d_u, d_v, ldo, d_s1, d_s2);
	return nrm_check_launch("k_nv_weights");
}

extern "C" int nrm_normvar_apply(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, const double* d_lnw, const double* d_wt,
								 const double* d_c, int64_t nc, int64_t ldc, const double* d_b, const double* d_scale, void* d_out, int out_dtype,
								 int64_t ldo, int32_t* d_flags, void* stream) {
	NRM_REQUIRE((y_dtype == NRM_F32 || y_dtype == NRM_F64) && (out_dtype == NRM_F32 || out_dtype == NRM_F64), "nrm_normvar_apply: bad dtype");
	NRM_REQUIRE(rows > 0 && n > 0 && ldy >= n && ldo >= n && nc > 0 && nc <= 64 && ldc >= n, "Unmatched gene or cell counts.");
	NRM_REQUIRE(d_y && d_lnw && d_wt && d_c && d_b && d_scale && d_out, "nrm_normvar_apply: null pointer");
	dim3 grid((unsigned)((rows + NV_R - 1) / NV_R));
	hipStream_t st = (hipStream_t)stream;
	const bool al = nv_aligned(d_y, y_dtype, ldy, d_lnw, d_c, ldc) && (uintptr_t)d_out % 16 == 0 && (ldo * (out_dtype == NRM_F64 ? 8 : 4)) % 16 == 0;
#define NV_LAUNCH2(TY, TO, AL) hipLaunchKernelGGL((k_nv_apply<TY, TO, AL>), grid, dim3(256), 0, st, (const TY*)d_y, rows, n, ldy, d_lnw, d_wt, d_c, (int)nc, ldc, d_b, d_scale, (TO*)d_out, ldo, d_flags)
#define NV_LAUNCH(TY, TO)        \
	do {                         \
		if (al)                  \
			NV_LAUNCH2(TY, TO, true);  \
		else                     \
			NV_LAUNCH2(TY, TO, false); \
	} while (0)
	if (y_dtype == NRM_F64 && out_dtype == NRM_F64) NV_LAUNCH(double, double);
	else if (y_dtype == NRM_F64) NV_LAUNCH(double, float);
	else if (out_dtype == NRM_F64) NV_LAUNCH(float, double);
	else NV_LAUNCH(float, float);
#undef NV_LAUNCH
#undef NV_LAUNCH2
	return nrm_check_launch("k_nv_apply");
}

// ---- round 5: the whole of normvar on the device, for up to NV_NC covariates ------------------------------------------------------------------
// The form above materialises U = e^2 and V = e^2 y (1.6 GB of fp64 at 5000 x 10 000) for two Gram launches, brings the per-gene moments to the
// host for 5000 pseudo-inverses and sends coefficients back: 16.9 ms per call at configs[1] size, 0.004 of the HBM rate (round-4 verdict).  Here:
//   k_nv_moments  a workgroup per gene reads the row ONCE and sums, per cell, e^2 C_c C_d (c <= d), e^2 y C_c, y e and (y e)^2 in registers;
//   k_nv_solve    a thread per gene: M_g^+ by the Jacobi iteration of nrm_jacobi.h (the host's code: the same integer ranks), b_g = M_g^+ a_g,
//                 the variance-keeping scale (norm.py:248-259);
//   k_nv_apply    (above) the second read of the row writes the result.
// Two reads of the matrix and one write: HBM-bound; nothing crosses PCIe.
#define NV_NC 8

#ifndef NV_G
#define NV_G 2  // genes per workgroup of the moments pass: they share the loads of the weights and the covariates, which every gene re-reads from L2 (480 KB per gene).
                // configs[1] size, moments + solve: one gene 0.203 ms (121 registers, four waves per SIMD), two 0.186 (175, two waves), four 0.266 (256, one wave)
#endif

template <typename T, int NC, bool ALIGNED, int G>
__global__ void __launch_bounds__(256) k_nv_moments(const T* __restrict__ y, int64_t rows, int64_t n, int64_t ldy, const double* __restrict__ lnw,
													 const double* __restrict__ wt, const double* __restrict__ c, int64_t ldc, double* __restrict__ mom) {
	constexpr int NP = NC * (NC + 1) / 2, NM = NP + NC + 2;
	__shared__ double sm[4][G * NM];
	__shared__ double tab[64];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	nv_exp_table(tab, tid);
	__syncthreads();
	const int64_t g0 = (int64_t)blockIdx.x * G;
	double ex[G];
	const T* row[G];
	double acc[G][NM];
#pragma unroll
	for (int u = 0; u < G; u++) {
		const int64_t g = g0 + u < rows ? g0 + u : rows - 1;  // (genes past the end repeat the last one: every load is issued, nothing of theirs is stored)
		ex[u] = wt[g];
		row[u] = y + g * ldy;
#pragma unroll
		for (int j = 0; j < NM; j++) acc[u][j] = 0.0;
	}
	auto cell = [&](int u, double lw, double yv, const double (&cv)[NC]) {
		const double e = ex[u] == 0.0 ? 1.0 : nv_exp(ex[u] * lw, tab);  // w**wt, exactly 1 for wt == 0 (norm.py:245)
		const double e2 = e * e, yp = yv * e;
		int j = 0;
#pragma unroll
		for (int q = 0; q < NC; q++) {
			const double t = e2 * cv[q];
#pragma unroll
			for (int d = q; d < NC; d++, j++) acc[u][j] = fma(t, cv[d], acc[u][j]);
		}
#pragma unroll
		for (int q = 0; q < NC; q++, j++) acc[u][j] = fma(e2 * yv, cv[q], acc[u][j]);
		acc[u][j] += yp;
		acc[u][j + 1] = fma(yp, yp, acc[u][j + 1]);
	};
	const int64_t n4 = n & ~(int64_t)3;
	for (int64_t k = (int64_t)tid * 4; k < n4; k += 1024) {  // four cells per thread: 2 + G + 2 NC sixteen-byte loads in flight before the first is used
		double lw[4], yv[G][4], cq[NC][4];
		nv_ld4<double, ALIGNED>(lnw + k, lw);
#pragma unroll
		for (int u = 0; u < G; u++) nv_ld4<T, ALIGNED>(row[u] + k, yv[u]);
#pragma unroll
		for (int q = 0; q < NC; q++) nv_ld4<double, ALIGNED>(c + (int64_t)q * ldc + k, cq[q]);
#pragma unroll
		for (int v = 0; v < 4; v++) {
			double cv[NC];
#pragma unroll
			for (int q = 0; q < NC; q++) cv[q] = cq[q][v];
#pragma unroll
			for (int u = 0; u < G; u++) cell(u, lw[v], yv[u][v], cv);
		}
	}
	for (int64_t k = n4 + tid; k < n; k += 256) {  // the last n % 4 cells
		double cv[NC];
#pragma unroll
		for (int q = 0; q < NC; q++) cv[q] = c[(int64_t)q * ldc + k];
#pragma unroll
		for (int u = 0; u < G; u++) cell(u, lnw[k], (double)row[u][k], cv);
	}
#pragma unroll
	for (int u = 0; u < G; u++)
#pragma unroll
		for (int j = 0; j < NM; j++) {
			const double t = nv_wave_sum(acc[u][j]);
			if (lane == 0) sm[wid][u * NM + j] = t;
		}
	__syncthreads();
	if (tid < G * NM && g0 + tid / NM < rows) mom[(g0 + tid / NM) * NM + tid % NM] = ((sm[0][tid] + sm[1][tid]) + sm[2][tid]) + sm[3][tid];
}

// mom (rows, NP + NC + 2) -> b (rows, NC), scale (rows), rank (rows); flags[0] += genes of rank 0
template <int NC>
__global__ void __launch_bounds__(64) k_nv_solve(const double* __restrict__ mom, int64_t rows, int64_t n, const double* __restrict__ wt, double tol, int keepvar,
												  double* __restrict__ b, double* __restrict__ scale, int64_t* __restrict__ rank, int32_t* __restrict__ flags) {
	constexpr int NP = NC * (NC + 1) / 2, NM = NP + NC + 2;
	const int64_t g = (int64_t)blockIdx.x * 64 + threadIdx.x;
	if (g >= rows) return;
	const double* mo = mom + g * NM;
	double m[NC * NC], inv[NC * NC];
	int j = 0;
	for (int q = 0; q < NC; q++)
		for (int d = q; d < NC; d++, j++) m[q * NC + d] = m[d * NC + q] = mo[j];
	int64_t rk = 0;
	nrm_small_pinv_one<NC>(m, NC, tol, inv, &rk);
	rank[g] = rk;
	if (rk <= 0) atomicAdd(&flags[0], 1);
	const double* a = mo + NP;
	double ab = 0.0;
	for (int q = 0; q < NC; q++) {
		double t = 0.0;
		for (int d = 0; d < NC; d++) t += inv[q * NC + d] * a[d];
		b[g * NC + q] = t;
		ab += a[q] * t;
	}
	double sc = 1.0;
	if (keepvar) {
		const double s1 = mo[NP + NC], s2 = mo[NP + NC + 1];
		const double mean = s1 / (double)n;
		const double dv = sqrt(fmax(s2 / (double)n - mean * mean, 0.0));  // norm.py:248-249
		const double dv2 = sqrt(fmax(s2 - ab, 0.0) / (double)n);          // |y' - P y'|^2 = |y'|^2 - a . b
		sc = pow(dv / dv2, wt[g]);                                          // norm.py:259
	}
	scale[g] = sc;
}

// every row of the matrix, the weights and the covariates starts on a 16-byte boundary: the vector-load instantiations
static bool nv_aligned(const void* d_y, int y_dtype, int64_t ldy, const double* d_lnw, const double* d_c, int64_t ldc) {
	return (uintptr_t)d_y % 16 == 0 && (ldy * (y_dtype == NRM_F64 ? 8 : 4)) % 16 == 0 && (uintptr_t)d_lnw % 16 == 0 && (uintptr_t)d_c % 16 == 0 && ldc % 2 == 0;
}

extern "C" int64_t nrm_normvar_device_covariates(void) { return NV_NC; }

// d_mom: rows x (nc (nc + 1) / 2 + nc + 2) doubles of scratch; d_b (rows, nc), d_scale (rows), d_rank (rows) int64; d_flags int32[4]: [0] += genes whose
// weighted covariates have rank 0 (norm.py:158-159 raises).  1 <= nc <= nrm_normvar_device_covariates().
extern "C" int nrm_normvar_solve(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, const double* d_lnw, const double* d_wt, const double* d_c, int64_t nc,
								 int64_t ldc, double tol, int keepvar, double* d_mom, double* d_b, double* d_scale, int64_t* d_rank, int32_t* d_flags, void* stream) {
	NRM_REQUIRE(y_dtype == NRM_F32 || y_dtype == NRM_F64, "nrm_normvar_solve: bad dtype");
	NRM_REQUIRE(rows > 0 && n > 0 && ldy >= n && nc >= 1 && nc <= NV_NC && ldc >= n && tol > 0, "nrm_normvar_solve: bad sizes (1 to %d covariates)", NV_NC);
	NRM_REQUIRE(d_y && d_lnw && d_wt && d_c && d_mom && d_b && d_scale && d_rank && d_flags, "nrm_normvar_solve: null pointer");
	hipStream_t st = (hipStream_t)stream;
	const bool al = nv_aligned(d_y, y_dtype, ldy, d_lnw, d_c, ldc);
#define NV_MOM(TY, NCV, AL) hipLaunchKernelGGL((k_nv_moments<TY, NCV, AL, NV_G>), dim3((unsigned)((rows + NV_G - 1) / NV_G)), dim3(256), 0, st, (const TY*)d_y, rows, n, ldy, d_lnw, d_wt, d_c, ldc, d_mom)
#define NV_GO(NCV)                                                                                                                                   \
	case NCV:                                                                                                                                        \
		if (y_dtype == NRM_F64) {                                                                                                                    \
			if (al) NV_MOM(double, NCV, true); else NV_MOM(double, NCV, false);                                                                      \
		} else {                                                                                                                                     \
			if (al) NV_MOM(float, NCV, true); else NV_MOM(float, NCV, false);                                                                        \
		}                                                                                                                                            \
		hipLaunchKernelGGL((k_nv_solve<NCV>), dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, d_mom, rows, n, d_wt, tol, keepvar, d_b, d_scale, d_rank, d_flags); \
		break;
	switch ((int)nc) {
		NV_GO(1) NV_GO(2) NV_GO(3) NV_GO(4) NV_GO(5) NV_GO(6) NV_GO(7) NV_GO(8)
	}
#undef NV_GO
#undef NV_MOM
	return nrm_check_launch("nrm_normvar_solve");
}

