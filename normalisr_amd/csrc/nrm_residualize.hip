// K1 + K4: residualise expression/design rows against the covariates once per row (the reference
// redoes this inside every tile: association.py:224-229) and take the row sums of squares
// (association.py:230-233) in the same pass.  HBM-bound.  A row is swept twice (products with the covariates, then residual -> digits):
// the second sweep is served from L2 only for short rows (configs[1]: 10 000 cells); from 50 000 cells up the rows a chip-full of
// workgroups holds between its sweeps (512 x 4 x 200 KB .. 4 MB) exceed L2 and the 256 MB Infinity Cache and every row is fetched from
// HBM twice (round-3 counters: 1.6 - 1.8x the algorithmic traffic).  Round 4 built the alternative -- rows resident in registers
// between the two phases (tools/experiments/nrm_residualize_res.hip) -- and measured why it does not win.
//
//   b_i  = (x_i C^T) dci          (association.py:226-227)
//   x~_i = x_i - b_i C            (association.py:228-229)
//   ss_i = sum_k x~_ik^2          (= n * variance, association.py:230)
//
// One workgroup (256 threads) owns RES_R consecutive rows so that every covariate element fetched
// from L2 is used RES_R times; covariates are swept in chunks of RES_CB to bound registers.
#include "nrm_k1.h"
#include <cstdlib>
#include <cstring>

#ifndef RES_R
#define RES_R 4
#endif
#define RES_CB 8
#ifndef K1_U1
#define K1_U1 2
#endif
#ifndef K1_U3
#define K1_U3 1
#endif

template <typename T>
__global__ void __launch_bounds__(256) k_residualize(const T* __restrict__ x, int64_t rows, int64_t n, int64_t ldx,
													  const double* __restrict__ c, int nc, int64_t ldc,
													  const double* __restrict__ dci, int active,
													  double* __restrict__ out, int64_t ldo, double* __restrict__ ss,
													  double* __restrict__ coef) {
	__shared__ double s_part[4][RES_R * RES_CB];
	extern __shared__ double s_dyn[];
	double* const ta = s_dyn;                    // [RES_R][nc]  x_i C^T
	double* const tb = s_dyn + (size_t)RES_R * nc;  // [RES_R][nc]  (x_i C^T) dci
	__shared__ double s_ss[4][RES_R];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int64_t row0 = (int64_t)blockIdx.x * RES_R;
	const T* xr[RES_R];
	bool live[RES_R];
#pragma unroll
	for (int r = 0; r < RES_R; r++) {
		live[r] = row0 + r < rows;
		xr[r] = x + (live[r] ? (row0 + r) : 0) * ldx;
	}

	if (active) {
		// phase A: a[r][cc] = sum_k x[r][k] C[cc][k]
		for (int c0 = 0; c0 < nc; c0 += RES_CB) {
			double acc[RES_R][RES_CB];
#pragma unroll
			for (int r = 0; r < RES_R; r++)
#pragma unroll
				for (int q = 0; q < RES_CB; q++) acc[r][q] = 0.0;
			for (int64_t k = tid; k < n; k += 256) {
				double xv[RES_R];
#pragma unroll
				for (int r = 0; r < RES_R; r++) xv[r] = live[r] ? (double)xr[r][k] : 0.0;
#pragma unroll
				for (int q = 0; q < RES_CB; q++) {
					double cv = (c0 + q < nc) ? c[(int64_t)(c0 + q) * ldc + k] : 0.0;
#pragma unroll
					for (int r = 0; r < RES_R; r++) acc[r][q] = fma(xv[r], cv, acc[r][q]);
				}
			}
#pragma unroll
			for (int r = 0; r < RES_R; r++)
#pragma unroll
				for (int q = 0; q < RES_CB; q++) {
					double v = wave_sum(acc[r][q]);
					if (lane == 0) s_part[wid][r * RES_CB + q] = v;
				}
			__syncthreads();
			if (tid < RES_R * RES_CB) {
				int r = tid / RES_CB, q = tid % RES_CB;
				if (c0 + q < nc) ta[r * nc + c0 + q] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
			}
			__syncthreads();
		}
		// b = a dci   (dci symmetric; association.py:226 applies dci on the left of dc@dx.T)
		for (int i = tid; i < RES_R * nc; i += 256) {
			int r = i / nc, q = i % nc;
			double v = 0.0;
			for (int e = 0; e < nc; e++) v = fma(dci[(int64_t)q * nc + e], ta[r * nc + e], v);
			tb[r * nc + q] = v;
			if (coef && live[r]) coef[(row0 + r) * nc + q] = v;
		}
		__syncthreads();
	}

	// phase B: residual, zero padding, sum of squares
	double sq[RES_R];
#pragma unroll
	for (int r = 0; r < RES_R; r++) sq[r] = 0.0;
	for (int64_t k = tid; k < ldo; k += 256) {
		double v[RES_R];
#pragma unroll
		for (int r = 0; r < RES_R; r++) v[r] = (live[r] && k < n) ? (double)xr[r][k] : 0.0;
		if (active && k < n) {
			for (int q = 0; q < nc; q++) {
				double cv = c[(int64_t)q * ldc + k];
#pragma unroll
				for (int r = 0; r < RES_R; r++) v[r] = fma(-tb[r * nc + q], cv, v[r]);
			}
		}
#pragma unroll
		for (int r = 0; r < RES_R; r++) {
			if (!live[r]) v[r] = 0.0;
			out[(row0 + r) * ldo + k] = v[r];
			sq[r] = fma(v[r], v[r], sq[r]);
		}
	}
#pragma unroll
	for (int r = 0; r < RES_R; r++) {
		double v = wave_sum(sq[r]);
		if (lane == 0) s_ss[wid][r] = v;
	}
	__syncthreads();
	if (tid < RES_R) ss[row0 + tid] = s_ss[0][tid] + s_ss[1][tid] + s_ss[2][tid] + s_ss[3][tid];
}

#ifndef K1_MINW
#define K1_MINW 1  // (waves per SIMD the register allocation must leave room for.  Round 6 A/B, profiles/r06_k1_occupancy.txt: 3 -> 168 registers + 12 spilled, C2 0.188 -> 0.213 ms;
                   // 4 -> 128 + 133 spilled, 0.44 ms: the 180 registers of two waves per SIMD are what the two sweeps need)
#endif
template <typename T, int CB, int NS, bool NT = false>
__global__ void __launch_bounds__(256, K1_MINW) k_residualize_v4(const T* __restrict__ x, int64_t rows, int64_t n, int64_t ldx,
														 const double* __restrict__ c, int nc, int64_t ldc,
														 const double* __restrict__ dci, int active, double* __restrict__ out,
														 int64_t ldo, double* __restrict__ ss, double* __restrict__ coef, QuantOut qo) {
	__shared__ double s_part[4][RES_R * CB];
	extern __shared__ double s_dyn[];
	double* const ta = s_dyn;                    // [RES_R][nc]  x_i C^T
	double* const tb = s_dyn + (size_t)RES_R * nc;  // [RES_R][nc]  (x_i C^T) dci
	__shared__ double s_ss[4][RES_R];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int64_t row0 = (int64_t)blockIdx.x * RES_R;
	const int64_t n4 = n & ~(int64_t)3;
	const T* xr[RES_R];
	bool live[RES_R];
#pragma unroll
	for (int r = 0; r < RES_R; r++) {
		live[r] = row0 + r < rows;
		xr[r] = x + (live[r] ? (row0 + r) : 0) * ldx;
	}
	double xmax[RES_R], xsq[RES_R];  // largest |x| and sum of squares of each row, gathered on the way (NS only)
#pragma unroll
	for (int r = 0; r < RES_R; r++) xmax[r] = xsq[r] = 0.0;
	if (active) {
		for (int c0 = 0; c0 < nc; c0 += CB) {
			double acc[RES_R][CB];
#pragma unroll
			for (int r = 0; r < RES_R; r++)
#pragma unroll
				for (int q = 0; q < CB; q++) acc[r][q] = 0.0;
#pragma unroll K1_U1
			for (int64_t k = (int64_t)tid * 4; k < n4; k += 1024) {
				double xv[RES_R][4];
#pragma unroll
				for (int r = 0; r < RES_R; r++) RowLoad<T, NT>::ld(xr[r] + k, xv[r]);
				if (NS && c0 == 0) {
#pragma unroll
					for (int r = 0; r < RES_R; r++)
#pragma unroll
						for (int i = 0; i < 4; i++) {
							xmax[r] = fmax(xmax[r], fabs(xv[r][i]));
							xsq[r] = fma(xv[r][i], xv[r][i], xsq[r]);
						}
				}
#pragma unroll
				for (int q = 0; q < CB; q++) {
					if (c0 + q < nc) {
						double cv[4];
						Vec4Load<double>::ld(c + (int64_t)(c0 + q) * ldc + k, cv);
#pragma unroll
						for (int r = 0; r < RES_R; r++)
#pragma unroll
							for (int i = 0; i < 4; i++) acc[r][q] = fma(xv[r][i], cv[i], acc[r][q]);
					}
				}
			}
			for (int64_t k = n4 + tid; k < n; k += 256) {  // tail cells when n % 4 != 0
				if (NS && c0 == 0) {
#pragma unroll
					for (int r = 0; r < RES_R; r++) {
						xmax[r] = fmax(xmax[r], fabs((double)xr[r][k]));
						xsq[r] = fma((double)xr[r][k], (double)xr[r][k], xsq[r]);
					}
				}
#pragma unroll
				for (int q = 0; q < CB; q++) {
					if (c0 + q < nc) {
						double cv = c[(int64_t)(c0 + q) * ldc + k];
#pragma unroll
						for (int r = 0; r < RES_R; r++) acc[r][q] = fma((double)xr[r][k], cv, acc[r][q]);
					}
				}
			}
#pragma unroll
			for (int r = 0; r < RES_R; r++)
#pragma unroll
				for (int q = 0; q < CB; q++) {
					double v = wave_sum(live[r] ? acc[r][q] : 0.0);
					if (lane == 0) s_part[wid][r * CB + q] = v;
				}
			__syncthreads();
			if (tid < RES_R * CB) {
				int r = tid / CB, q = tid % CB;
				if (c0 + q < nc) ta[r * nc + c0 + q] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
			}
			__syncthreads();
		}
		for (int i = tid; i < RES_R * nc; i += 256) {
			int r = i / nc, q = i % nc;
			double v = 0.0;
			for (int e = 0; e < nc; e++) v = fma(dci[(int64_t)q * nc + e], ta[r * nc + e], v);
			tb[r * nc + q] = v;
			if (coef && live[r]) coef[(row0 + r) * nc + q] = v;
		}
		__syncthreads();
	}
	// residual of 4 consecutive cells k..k+3 of the workgroup's rows (zeros past the end of the row and for padding rows)
	auto residual4 = [&](int64_t k, double (&v)[RES_R][4]) {
		if (k < n4) {
#pragma unroll
			for (int r = 0; r < RES_R; r++) RowLoad<T, NT>::ld(xr[r] + k, v[r]);
			if (active) {
				for (int q = 0; q < nc; q++) {
					double cv[4];
					Vec4Load<double>::ld(c + (int64_t)q * ldc + k, cv);
#pragma unroll
					for (int r = 0; r < RES_R; r++)
#pragma unroll
						for (int i = 0; i < 4; i++) v[r][i] = fma(-tb[r * nc + q], cv[i], v[r][i]);
				}
			}
		} else {
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int64_t kk = k + i;
#pragma unroll
				for (int r = 0; r < RES_R; r++) v[r][i] = (kk < n) ? (double)xr[r][kk] : 0.0;
				if (active && kk < n) {
					for (int q = 0; q < nc; q++) {
						double cv = c[(int64_t)q * ldc + kk];
#pragma unroll
						for (int r = 0; r < RES_R; r++) v[r][i] = fma(-tb[r * nc + q], cv, v[r][i]);
					}
				}
			}
		}
#pragma unroll
		for (int r = 0; r < RES_R; r++)
			if (!live[r]) v[r][0] = v[r][1] = v[r][2] = v[r][3] = 0.0;
	};
	// Fixed-point scale of each row (NS only): 2^e >= the largest |residual|.  With covariates a bound does when it is tight: |x - b C|
	// <= max|x| + sum_c |b_c| max|C_c| (max|x| was gathered in the first sweep, max|C_c| comes from the caller) saves a sweep over the
	// row.  It is accepted when it lies within RES_LOOSE of the residuals' rms (|x|^2 - a.b from the same sweep); otherwise, and
	// without covariates or without the caller's maxima, the residuals are swept once more for their true maximum.
	__shared__ double s_mx[4][RES_R], s_sq[4][RES_R];
	__shared__ int s_sh[RES_R], s_loose;
	int sh[RES_R];
	if (NS) {
		constexpr int B = 8 * NS - 2;
		const bool bounded = active && qo.cmax != nullptr;
		auto reduce_max = [&]() {
#pragma unroll
			for (int r = 0; r < RES_R; r++) {
				double v = xmax[r], q = wave_sum(xsq[r]);
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
				if (lane == 0) {
					s_mx[wid][r] = v;
					s_sq[wid][r] = q;
				}
			}
		};
		if (tid == 0) s_loose = 0;
		if (bounded) {
			reduce_max();
			__syncthreads();
			if (tid < RES_R) {
				double m = fmax(fmax(s_mx[0][tid], s_mx[1][tid]), fmax(s_mx[2][tid], s_mx[3][tid]));
				const double sq = (s_sq[0][tid] + s_sq[1][tid]) + (s_sq[2][tid] + s_sq[3][tid]);
				double proj = 0.0;
				for (int q = 0; q < nc; q++) {
					m = fma(fabs(tb[tid * nc + q]), qo.cmax[q], m);
					proj = fma(ta[tid * nc + q], tb[tid * nc + q], proj);
				}
				const double est = sq - proj;  // |x~|^2 up to cancellation: trusted only while it is a fair share of |x|^2
				if (live[tid] && !(est > 1e-8 * sq && m * m * (double)n <= (RES_LOOSE * RES_LOOSE) * est)) s_loose = 1;
				s_mx[0][tid] = m;
			}
			__syncthreads();
		}
		const bool sweep = !bounded || s_loose != 0;  // (one decision for the workgroup's rows)
		if (sweep) {
			__syncthreads();
#pragma unroll
			for (int r = 0; r < RES_R; r++) xmax[r] = xsq[r] = 0.0;
			for (int64_t k = (int64_t)tid * 4; k < ((n + 3) & ~(int64_t)3); k += 1024) {
				double v[RES_R][4];
				residual4(k, v);
#pragma unroll
				for (int r = 0; r < RES_R; r++)
#pragma unroll
					for (int i = 0; i < 4; i++) xmax[r] = fmax(xmax[r], fabs(v[r][i]));
			}
			reduce_max();
			__syncthreads();
			if (tid < RES_R) s_mx[0][tid] = fmax(fmax(s_mx[0][tid], s_mx[1][tid]), fmax(s_mx[2][tid], s_mx[3][tid]));
			__syncthreads();
		}
		if (tid < RES_R) {
			const double m = s_mx[0][tid];
			int e = 0;
			if (m > 0.0 && m < INFINITY) (void)frexp(m, &e);  // m = f 2^e, f in [0.5, 1): every |residual| < 2^e
			s_sh[tid] = e - B;
			qo.exps[row0 + tid] = e - B;
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < RES_R; r++) sh[r] = s_sh[r];
	}
	char* qrow[RES_R];
	int flip[RES_R];
#pragma unroll
	for (int r = 0; r < RES_R; r++) {
		const int64_t row = row0 + r;
		const int rr = (int)(row & 31);
		qrow[r] = NS ? qo.q + ((row >> 5) * qo.cks) * 1024 + (2 * rr) * 16 : nullptr;
		flip[r] = (rr >> 3) & 1;
	}
	// last sweep (the rows are in L2 / MALL by now): residual, zero padding, sum of squares and -- NS -- the residual rounded
	// once to (8 NS - 2)-bit fixed point and cut into NS balanced base-256 digits (layout: nrm_gram_i8.hip)
	double sq[RES_R];
#pragma unroll
	for (int r = 0; r < RES_R; r++) sq[r] = 0.0;
	// digit sums and sums of squares of the planes that have dropped products (s <= NS - 2): the row record of nrm_fix.h
	constexpr int NP = NS >= 2 ? NS - 1 : 1;
	int dsum[RES_R][NP];
	unsigned dsq[RES_R][NP];
#pragma unroll
	for (int r = 0; r < RES_R; r++)
#pragma unroll
		for (int s = 0; s < NP; s++) {
			dsum[r][s] = 0;
			dsq[r][s] = 0u;
		}
	const int64_t kres = out ? ldo : ((n + 3) & ~(int64_t)3), kq = NS ? qo.nks * 32 : 0;
	// The rows of the NEXT step are requested before this step's are used (raw values, 4 registers per row): without that a step waits
	// for its own HBM loads -- 16 KB in flight per workgroup, ~3 TB/s on the chip (round-4 measurement: the same serialisation cost
	// binnet 2.5x).  The covariates come from L2 inside the step.
	T xnext[RES_R][4];
	auto request = [&](int64_t k, T (&raw)[RES_R][4]) {
		if (k < n4) {
#pragma unroll
			for (int r = 0; r < RES_R; r++) RawLoad<T, NT>::ld(xr[r] + k, raw[r]);
		} else {
#pragma unroll
			for (int r = 0; r < RES_R; r++)
#pragma unroll
				for (int i = 0; i < 4; i++) raw[r][i] = (k + i < n) ? xr[r][k + i] : (T)0;
		}
	};
	// residual of the 4 cells from raw values already in registers
	auto residual_of = [&](int64_t k, const T (&raw)[RES_R][4], double (&v)[RES_R][4]) {
#pragma unroll
		for (int r = 0; r < RES_R; r++)
#pragma unroll
			for (int i = 0; i < 4; i++) v[r][i] = (double)raw[r][i];
		if (active && k < n) {
			if (k < n4) {
				for (int q = 0; q < nc; q++) {
					double cv[4];
					Vec4Load<double>::ld(c + (int64_t)q * ldc + k, cv);
#pragma unroll
					for (int r = 0; r < RES_R; r++)
#pragma unroll
						for (int i = 0; i < 4; i++) v[r][i] = fma(-tb[r * nc + q], cv[i], v[r][i]);
				}
			} else {
				for (int q = 0; q < nc; q++)
#pragma unroll
					for (int i = 0; i < 4; i++) {
						const double cv = (k + i < n) ? c[(int64_t)q * ldc + k + i] : 0.0;
#pragma unroll
						for (int r = 0; r < RES_R; r++) v[r][i] = fma(-tb[r * nc + q], cv, v[r][i]);
					}
			}
		}
#pragma unroll
		for (int r = 0; r < RES_R; r++)
			if (!live[r]) v[r][0] = v[r][1] = v[r][2] = v[r][3] = 0.0;
	};
	// (fp32 rows: 3.70 -> 3.27 ms on 16 000 x 50 000 with 5 covariates; fp64 rows -- twice the registers for the same cells -- ran 4 % slower
	// with it on the configs[4] slice, so they keep requesting their rows inside the step)
	constexpr bool AHEAD = sizeof(T) == 4;
	const int64_t kend = kres > kq ? kres : kq;
	if (AHEAD && (int64_t)tid * 4 < kend) request((int64_t)tid * 4, xnext);
#pragma unroll K1_U3
	for (int64_t k = (int64_t)tid * 4; k < kend; k += 1024) {
		T xcur[RES_R][4];
		if (AHEAD) {
#pragma unroll
			for (int r = 0; r < RES_R; r++)
#pragma unroll
				for (int i = 0; i < 4; i++) xcur[r][i] = xnext[r][i];
			if (k + 1024 < kend) request(k + 1024, xnext);
		} else
			request(k, xcur);
		double v[RES_R][4];
		residual_of(k, xcur, v);
#pragma unroll
		for (int r = 0; r < RES_R; r++) {
			if (out && k < ldo) {
				double* o = out + (row0 + r) * ldo + k;
				*reinterpret_cast<double2*>(o) = make_double2(v[r][0], v[r][1]);
				*reinterpret_cast<double2*>(o + 2) = make_double2(v[r][2], v[r][3]);
			}
#pragma unroll
			for (int i = 0; i < 4; i++) sq[r] = fma(v[r][i], v[r][i], sq[r]);
		}
		if (NS && k < kq) {
			const int kk = (int)(k & 31);
			const int ks_all = (int)(k >> 5), chunk = ks_all / (int)qo.cks;
			const int64_t ks = (int64_t)(ks_all - chunk * (int)qo.cks) + chunk * (qo.chunk_bytes >> 10);  // in KB images from q
#pragma unroll
			for (int r = 0; r < RES_R; r++) {
				unsigned w[NS ? NS : 1];
				nrm_digits4<(NS ? NS : 1)>(v[r], sh[r], w);
				char* dst = qrow[r] + ks * 1024 + (((kk >> 4) ^ flip[r]) << 4) + (kk & 15);
#pragma unroll
				for (int s = 0; s < (NS ? NS : 1); s++) *reinterpret_cast<unsigned*>(dst + s * qo.plane_bytes) = w[s];
#pragma unroll
				for (int s = 0; s < NP; s++) {
					dsum[r][s] = __builtin_amdgcn_sdot4((int)w[s], 0x01010101, dsum[r][s], false);
					dsq[r][s] = (unsigned)__builtin_amdgcn_sdot4((int)w[s], (int)w[s], (int)dsq[r][s], false);
				}
			}
		}
	}
#pragma unroll
	for (int r = 0; r < RES_R; r++) {
		double v = wave_sum(sq[r]);
		if (lane == 0) s_ss[wid][r] = v;
	}
	__shared__ long long s_ds[4][RES_R][NP], s_dq[4][RES_R][NP];
	if (NS && qo.fix) {
#pragma unroll
		for (int r = 0; r < RES_R; r++)
#pragma unroll
			for (int s = 0; s < NP; s++) {
				const long long a = wave_sum_i32(dsum[r][s]), b = wave_sum_u32(dsq[r][s]);
				if (lane == 0) {
					s_ds[wid][r][s] = a;
					s_dq[wid][r][s] = b;
				}
			}
	}
	__syncthreads();
	if (tid < RES_R) {
		const double ssr = s_ss[0][tid] + s_ss[1][tid] + s_ss[2][tid] + s_ss[3][tid];
		ss[row0 + tid] = ssr;
		if (NS && qo.fix) {
			double S[5] = {0, 0, 0, 0, 0}, Q[5] = {0, 0, 0, 0, 0};
#pragma unroll
			for (int s = 0; s < NP; s++) {
				S[s] = (double)((s_ds[0][tid][s] + s_ds[1][tid][s]) + (s_ds[2][tid][s] + s_ds[3][tid][s]));
				Q[s] = (double)((s_dq[0][tid][s] + s_dq[1][tid][s]) + (s_dq[2][tid][s] + s_dq[3][tid][s]));
			}
			nrm_fix_record<NS>(qo.fix + (row0 + tid) * NRM_FIX_STRIDE, S, Q, s_sh[tid], ssr, (double)n);
		}
	}
}

template <typename T>
static void launch_residualize(bool vec, const T* x, int64_t rows, int64_t n, int64_t ldx, const double* c, int nc, int64_t ldc,
							   const double* dci, int active, double* out, int64_t ldo, int64_t rows_pad, double* ss, double* coef,
							   int nslices, QuantOut qo, hipStream_t st) {
	dim3 grid((unsigned)(rows_pad / RES_R));
	const size_t lds = (size_t)2 * RES_R * (nc > 0 ? nc : 1) * sizeof(double);
	// input rows loaded non-temporally (nrm_k1.h): fp32 rows against at least 8 MB of covariates, which then stay in L2
	const bool stream_rows = active && sizeof(T) == 4 && (int64_t)nc * n * 8 >= (8 << 20);
	auto go = [&](auto kern, auto... extra) {
		if (lds > 48 * 1024)  // beyond the default dynamic-LDS window (more than 768 covariates)
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, rows, n, ldx, c, nc, ldc, dci, active, out, ldo, ss, coef, extra...);
	};
	if (!vec)
		go(k_residualize<T>);
	else if (nslices == 6)
		nc <= 4 ? go(k_residualize_v4<T, 4, 6>, qo) : (stream_rows ? go(k_residualize_v4<T, 8, 6, true>, qo) : go(k_residualize_v4<T, 8, 6>, qo));
	else if (nslices == 5)
		nc <= 4 ? go(k_residualize_v4<T, 4, 5>, qo) : (stream_rows ? go(k_residualize_v4<T, 8, 5, true>, qo) : go(k_residualize_v4<T, 8, 5>, qo));
	else
		nc <= 4 ? go(k_residualize_v4<T, 4, 0>, qo) : go(k_residualize_v4<T, 8, 0>, qo);
}

static int residualize_impl(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx, const double* d_c, int64_t nc, int64_t ldc,
							const double* d_dci, int rank, double* d_out, int64_t ldo, int64_t rows_pad, double* d_ss, double* d_coef,
							int nslices, void* d_q, int32_t* d_exp, int64_t plane_pitch, int64_t chunk_ksteps, const double* d_cmax, double* d_fix,
							void* stream) {
	NRM_REQUIRE(x_dtype == NRM_F32 || x_dtype == NRM_F64, "nrm_residualize: bad dtype");
	NRM_REQUIRE(rows >= 0 && n > 0 && ldx >= n, "Incorrect dx/dy/dc size.");
	NRM_REQUIRE(nc >= 0 && nc <= RES_NC_MAX, "nrm_residualize: at most %d covariates supported", RES_NC_MAX);
	NRM_REQUIRE(rank >= 0 && rank <= nc, "dcr higher than covariate dimension.");
	NRM_REQUIRE(rows_pad >= rows && rows_pad % RES_R == 0, "nrm_residualize: rows_pad must cover rows and be a multiple of %d", RES_R);
	NRM_REQUIRE(d_out ? ldo >= n : nslices != 0, "nrm_residualize: output pitch smaller than cell count (or no output requested)");
	NRM_REQUIRE(d_ss, "nrm_residualize: null output");
	if (rows_pad == 0) return NRM_OK;
	NRM_REQUIRE(d_x || rows == 0, "nrm_residualize: null input");
	int active = (rank > 0 && nc > 0) ? 1 : 0;
	NRM_REQUIRE(!active || (d_c && d_dci && ldc >= n), "Unmatching dx/dy/dc dimensions.");
	const int64_t xa = 16 / (x_dtype == NRM_F64 ? 8 : 4);
	const bool vec = (ldx % xa == 0) && ((uintptr_t)d_x % 16 == 0) && (!d_out || ((ldo % 4 == 0) && ((uintptr_t)d_out % 16 == 0))) &&
					 (!active || (ldc % 2 == 0 && (uintptr_t)d_c % 16 == 0));
	QuantOut qo = {nullptr, 0, 0, 0, 0, nullptr, d_cmax, d_fix};
	if (nslices) {
		NRM_REQUIRE(nslices == 5 || nslices == 6, "nrm_residualize_q: 5 or 6 slices");
		NRM_REQUIRE(n < (1 << 22), "nrm_residualize_q: the integer engine takes rows of fewer than 2^22 cells");
		NRM_REQUIRE(vec, "nrm_residualize_q: needs 16-byte aligned rows (use nrm_residualize + nrm_quantize_rows otherwise)");
		NRM_REQUIRE(rows_pad % 128 == 0 && d_q && d_exp && (uintptr_t)d_q % 16 == 0, "nrm_residualize_q: rows_pad %% 128 == 0 and digit buffers required");
		const int64_t k_pad = (n + 15) / 16 * 16;
		qo.nks = (k_pad + 31) / 32;
		qo.cks = qo.nks;
		if (chunk_ksteps > 0) {  // cell chunks of chunk_ksteps k-steps, each a dense operand of its own; the last one zero padded
			NRM_REQUIRE(plane_pitch == 0, "nrm_residualize_q_chunked: chunks are dense");
			qo.cks = chunk_ksteps;
			qo.nks = (qo.nks + qo.cks - 1) / qo.cks * qo.cks;
		}
		qo.plane_bytes = (rows_pad / 32) * qo.cks * 1024;
		qo.chunk_bytes = nslices * qo.plane_bytes;
		if (plane_pitch) {  // these rows are a block of a larger quantised matrix
			NRM_REQUIRE(plane_pitch >= qo.plane_bytes && plane_pitch % 1024 == 0, "nrm_residualize_q: plane pitch smaller than the block");
			qo.plane_bytes = plane_pitch;
		}
		qo.q = (char*)d_q;
		qo.exps = d_exp;
	}
	if (x_dtype == NRM_F64)
		launch_residualize<double>(vec, (const double*)d_x, rows, n, ldx, d_c, (int)nc, ldc, d_dci, active, d_out, ldo, rows_pad, d_ss, d_coef,
								   nslices, qo, (hipStream_t)stream);
	else
		launch_residualize<float>(vec, (const float*)d_x, rows, n, ldx, d_c, (int)nc, ldc, d_dci, active, d_out, ldo, rows_pad, d_ss, d_coef,
								  nslices, qo, (hipStream_t)stream);
	return nrm_check_launch("k_residualize");
}

extern "C" int nrm_residualize(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx, const double* d_c,
							   int64_t nc, int64_t ldc, const double* d_dci, int rank, double* d_out, int64_t ldo,
							   int64_t rows_pad, double* d_ss, double* d_coef, void* stream) {
	NRM_REQUIRE(d_out != nullptr, "nrm_residualize: null output");
	return residualize_impl(d_x, x_dtype, rows, n, ldx, d_c, nc, ldc, d_dci, rank, d_out, ldo, rows_pad, d_ss, d_coef, 0, nullptr, nullptr, 0, 0, nullptr,
							nullptr, stream);
}

extern "C" int nrm_residualize_q(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx, const double* d_c, int64_t nc,
								 int64_t ldc, const double* d_dci, int rank, double* d_out, int64_t ldo, int64_t rows_pad, double* d_ss,
								 double* d_coef, int nslices, void* d_q, int32_t* d_exp, int64_t plane_pitch_bytes, const double* d_cmax,
								 double* d_fix, void* stream) {
	return residualize_impl(d_x, x_dtype, rows, n, ldx, d_c, nc, ldc, d_dci, rank, d_out, ldo, rows_pad, d_ss, d_coef, nslices, d_q, d_exp,
							plane_pitch_bytes, 0, d_cmax, d_fix, stream);
}

// The same with the digit planes cut along the cells into chunks of chunk_ksteps * 32 cells: chunk c is a dense quantised
// operand of its own (nslices planes of rows_pad / 32 * chunk_ksteps KB) at d_q + c * nslices * plane bytes, all chunks share
// the row exponents.  The sharded coex path sends the chunks one after another and contracts each as it lands
// (nrm_gram_i8_chunk).  d_q: ceil(ceil(k_pad / 32) / chunk_ksteps) * nrm_quant_bytes(rows_pad, 32 * chunk_ksteps, nslices) bytes.
extern "C" int nrm_residualize_q_chunked(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx, const double* d_c, int64_t nc,
										 int64_t ldc, const double* d_dci, int rank, int64_t rows_pad, double* d_ss, int nslices, void* d_q,
										 int32_t* d_exp, int64_t chunk_ksteps, const double* d_cmax, double* d_fix, void* stream) {
	NRM_REQUIRE(chunk_ksteps > 0 && chunk_ksteps < (1 << 24), "nrm_residualize_q_chunked: bad chunk size");
	return residualize_impl(d_x, x_dtype, rows, n, ldx, d_c, nc, ldc, d_dci, rank, nullptr, 0, rows_pad, d_ss, nullptr, nslices, d_q, d_exp, 0,
							chunk_ksteps, d_cmax, d_fix, stream);
}

// Few design rows (streaming de path): the work is spread along the CELLS instead of the rows.  The OLS
// products a = x C^T come from nrm_gram_skinny (G matrix, 32 columns); every workgroup rebuilds b = a dci in
// LDS (tiny) and applies x~ = x - b C to its 1024-cell slab for all rows; sums of squares by atomics.
#define RW_ROWS 32
template <typename T>
__global__ void __launch_bounds__(256) k_residualize_wide(const T* __restrict__ x, int rows, int64_t n, int64_t ldx,
														   const double* __restrict__ c, int nc, int64_t ldc,
														   const double* __restrict__ ga /* (rows, 32) */, const double* __restrict__ dci,
														   int active, double* __restrict__ out, int64_t ldo, double* __restrict__ ss_part,
														   double* __restrict__ coef, int const_last) {
	__shared__ double s_b[RW_ROWS][RW_ROWS];
	__shared__ double s_w[4], s_r[4];
	const int tid = threadIdx.x, lane = tid & 63;
	if (active) {
		for (int i = tid; i < rows * nc; i += 256) {
			const int r = i / nc, q = i % nc;
			double v = 0.0;
			for (int e = 0; e < nc; e++) v = fma(dci[(int64_t)q * nc + e], ga[r * 32 + ((const_last && e == nc - 1) ? 31 : e)], v);
			s_b[r][q] = v;
			if (coef && blockIdx.x == 0) coef[r * nc + q] = v;
		}
		__syncthreads();
	}
	const int64_t k0 = (int64_t)blockIdx.x * 1024;
	for (int r = 0; r < rows; r++) {
		double sq = 0.0, raw = 0.0;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int64_t k = k0 + tid + 256 * j;
			if (k >= ldo) continue;
			double v = (k < n) ? (double)x[(int64_t)r * ldx + k] : 0.0;
			raw = fma(v, v, raw);
			if (active && k < n)
				for (int q = 0; q < nc; q++) v = fma(-s_b[r][q], c[(int64_t)q * ldc + k], v);
			out[(int64_t)r * ldo + k] = v;
			sq = fma(v, v, sq);
		}
		// per-block partial sums (no atomics: k_rw_sum adds them in block order, bitwise reproducible); the raw rows' squares ride along (k_rw_sum)
		sq = wave_sum(sq);
		raw = wave_sum(raw);
		__syncthreads();
		if (lane == 0) {
			s_w[tid >> 6] = sq;
			s_r[tid >> 6] = raw;
		}
		__syncthreads();
		if (tid == 0) {
			ss_part[(int64_t)blockIdx.x * RW_ROWS + r] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
			ss_part[((int64_t)gridDim.x + blockIdx.x) * RW_ROWS + r] = s_r[0] + s_r[1] + s_r[2] + s_r[3];
		}
	}
}

// one workgroup per row: the per-block partial sums are added in a fixed (strided, then tree) order -- bitwise reproducible
// A row the covariates explain to twenty digits (|x~|^2 < 1e-22 |x|^2: a constant design row beside an intercept, a copy of a covariate) is explained
// exactly: its residual is rounding noise, NOT orthogonal to the covariates, and the streaming kernel's y~ . x~ = y . x~ would make an R^2 of millions of it.
// The row of `out` is cleared and its sum of squares is 0 (the variance 0 -> 1 rule then gives P = 1: what exact arithmetic gives the reference).
__global__ void __launch_bounds__(256) k_rw_sum(const double* __restrict__ part, int nblocks, int rows, double* __restrict__ ss, double* __restrict__ out, int64_t ldo) {
	__shared__ double s_w[4], s_r[4];
	__shared__ int s_clear;
	const int r = blockIdx.x;
	double acc = 0.0, raw = 0.0;
	for (int b = threadIdx.x; b < nblocks; b += 256) {
		acc += part[(int64_t)b * RW_ROWS + r];
		raw += part[((int64_t)nblocks + b) * RW_ROWS + r];
	}
	acc = wave_sum(acc);
	raw = wave_sum(raw);
	if ((threadIdx.x & 63) == 0) {
		s_w[threadIdx.x >> 6] = acc;
		s_r[threadIdx.x >> 6] = raw;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		const double a = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]), q = (s_r[0] + s_r[1]) + (s_r[2] + s_r[3]);
		const int clear = a < 1e-22 * q;
		ss[r] = clear ? 0.0 : a;
		s_clear = clear;
	}
	__syncthreads();
	if (s_clear)
		for (int64_t k = threadIdx.x; k < ldo; k += 256) out[(int64_t)r * ldo + k] = 0.0;
}

// The OLS products a = x C^T of a few design rows, spread along the cells like k_residualize_wide: every workgroup takes 1024 cells of
// all rows and covariates, k_xc_sum adds the per-block partial sums in a fixed order (bitwise reproducible, no atomics).  Output in the
// layout k_residualize_wide reads: ga[r * 32 + c], the last covariate in column 31 when const_last.  (Until late in round 3 these
// came from nrm_gram_skinny on a 256-row tile holding the 1 - 31 design rows: one tile cut into 256 stream-K pieces, whose fix-up
// then added 256 slabs in a chain -- 36 + 48 us per step of configs[2], against 8 + 4 here.)
template <typename T>
__global__ void __launch_bounds__(256) k_xc_partial(const T* __restrict__ x, int rows, int64_t n, int64_t ldx, const double* __restrict__ c, int nc,
													 int64_t ldc, double* __restrict__ part) {
	__shared__ double s_w[4][RW_ROWS];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int64_t k0 = (int64_t)blockIdx.x * 1024;
	for (int r = 0; r < rows; r++) {
		double xv[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int64_t k = k0 + tid + 256 * j;
			xv[j] = k < n ? (double)x[(int64_t)r * ldx + k] : 0.0;
		}
		for (int q0 = 0; q0 < nc; q0 += 8) {  // 8 covariates at a time: 32 independent loads in flight per thread
			double cv[8][4];
#pragma unroll
			for (int i = 0; i < 8; i++)
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const int64_t k = k0 + tid + 256 * j;
					cv[i][j] = (q0 + i < nc && k < n) ? c[(int64_t)(q0 + i) * ldc + k] : 0.0;
				}
#pragma unroll
			for (int i = 0; i < 8; i++) {
				double acc = 0.0;
#pragma unroll
				for (int j = 0; j < 4; j++) acc = fma(xv[j], cv[i][j], acc);
				acc = wave_sum(acc);
				if (lane == 0 && q0 + i < nc) s_w[wid][q0 + i] = acc;
			}
		}
		__syncthreads();
		if (tid < nc) part[((int64_t)blockIdx.x * rows + r) * RW_ROWS + tid] = (s_w[0][tid] + s_w[1][tid]) + (s_w[2][tid] + s_w[3][tid]);
		__syncthreads();
	}
}

// one workgroup per design row: thread (g, q) adds the partial sums of blocks g, g + 8, ... of covariate q, the 8 groups are added in order
__global__ void __launch_bounds__(256) k_xc_sum(const double* __restrict__ part, int nblocks, int rows, int nc, double* __restrict__ ga, int const_last) {
	__shared__ double s_g[8][RW_ROWS];
	const int r = blockIdx.x, q = threadIdx.x & 31, g = threadIdx.x >> 5;
	double acc = 0.0;
	if (q < nc)
		for (int b = g; b < nblocks; b += 8) acc += part[((int64_t)b * rows + r) * RW_ROWS + q];
	s_g[g][q] = acc;
	__syncthreads();
	if (threadIdx.x < nc) {
		double v = s_g[0][q];
#pragma unroll
		for (int i = 1; i < 8; i++) v += s_g[i][q];
		ga[r * 32 + ((const_last && q == nc - 1) ? 31 : q)] = v;
	}
}

extern "C" int64_t nrm_design_products_workspace_doubles(int64_t rows, int64_t n) { return rows * RW_ROWS * ((n + 1023) / 1024); }

extern "C" int nrm_design_products(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx, const double* d_c, int64_t nc, int64_t ldc,
								   double* d_ga, double* d_work, int const_last, void* stream) {
	NRM_REQUIRE(x_dtype == NRM_F32 || x_dtype == NRM_F64, "nrm_design_products: bad dtype");
	NRM_REQUIRE(rows > 0 && rows <= RW_ROWS && nc > 0 && nc <= RW_ROWS, "nrm_design_products: 1 to %d rows and covariates", RW_ROWS);
	NRM_REQUIRE(n > 0 && ldx >= n && ldc >= n && d_x && d_c && d_ga && d_work, "Incorrect dx/dy/dc size.");
	hipStream_t st = (hipStream_t)stream;
	const dim3 grid((unsigned)((n + 1023) / 1024));
	if (x_dtype == NRM_F64)
		hipLaunchKernelGGL(k_xc_partial<double>, grid, dim3(256), 0, st, (const double*)d_x, (int)rows, n, ldx, d_c, (int)nc, ldc, d_work);
	else
		hipLaunchKernelGGL(k_xc_partial<float>, grid, dim3(256), 0, st, (const float*)d_x, (int)rows, n, ldx, d_c, (int)nc, ldc, d_work);
	hipLaunchKernelGGL(k_xc_sum, dim3((unsigned)rows), dim3(256), 0, st, d_work, (int)grid.x, (int)rows, (int)nc, d_ga, const_last);
	return nrm_check_launch("k_xc_partial");
}

extern "C" int nrm_residualize_wide(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx, const double* d_c, int64_t nc,
									int64_t ldc, const double* d_ga, const double* d_dci, int rank, double* d_out, int64_t ldo,
									double* d_ss, double* d_coef, double* d_work, int const_last, void* stream) {
	NRM_REQUIRE(x_dtype == NRM_F32 || x_dtype == NRM_F64, "nrm_residualize_wide: bad dtype");
	NRM_REQUIRE(rows > 0 && rows <= RW_ROWS && nc >= 0 && nc <= RW_ROWS, "nrm_residualize_wide: at most %d rows and covariates", RW_ROWS);
	NRM_REQUIRE(n > 0 && ldx >= n && ldo >= n && d_x && d_out && d_ss, "Incorrect dx/dy/dc size.");
	const int active = (rank > 0 && nc > 0) ? 1 : 0;
	NRM_REQUIRE(!active || (d_c && d_ga && d_dci && ldc >= n), "Unmatching dx/dy/dc dimensions.");
	hipStream_t st = (hipStream_t)stream;
	NRM_REQUIRE(d_work != nullptr, "nrm_residualize_wide: d_work must hold 64 * ceil(ldo / 1024) doubles");
	dim3 grid((unsigned)((ldo + 1023) / 1024));
	if (x_dtype == NRM_F64)
		hipLaunchKernelGGL(k_residualize_wide<double>, grid, dim3(256), 0, st, (const double*)d_x, (int)rows, n, ldx, d_c, (int)nc, ldc, d_ga,
						   d_dci, active, d_out, ldo, d_work, d_coef, const_last);
	else
		hipLaunchKernelGGL(k_residualize_wide<float>, grid, dim3(256), 0, st, (const float*)d_x, (int)rows, n, ldx, d_c, (int)nc, ldc, d_ga,
						   d_dci, active, d_out, ldo, d_work, d_coef, const_last);
	hipLaunchKernelGGL(k_rw_sum, dim3((unsigned)rows), dim3(256), 0, st, d_work, (int)grid.x, (int)rows, d_ss, d_out, ldo);
	return nrm_check_launch("k_residualize_wide");
}
