// K2s: de with few design rows (case-control DE, config C3: 1 x 20k x 100k cells) is HBM-bound -- every
// expression value is needed once.  This kernel streams the RAW expression rows (fp32 or fp64, as handed
// over by numpy, never materialised as fp64 residuals) exactly once and contracts them on the fp64 matrix
// cores against Z = [C; X~] (covariates and the already-residualised design rows, <= 32 rows, fp64):
//
//     G[y, :] = sum_k Y[y,k] Z[:,k]        (ny, 32)      ss[y] = sum_k Y[y,k]^2
//
// from which the per-pair sweep (k_de_small_sweep) recovers, without ever forming y~:
//     y~ . x~ = y . x~                      (x~ is orthogonal to C)           association.py:234
//     |y~|^2  = |y|^2 - (y C^T) dci (C y^T)                                   association.py:229-230
// Algorithmic HBM bytes: itemsize * n per expression row (+ Z from L2).  Geometry: workgroup = 128 rows x
// 32 Z-rows, 4 waves stacked along the rows (each 32 x 32 = 2 x 2 MFMA tiles), K slabs of 32 cells staged
// global -> registers -> LDS (fp32 converted on the way); persistent DP + stream-K schedule as in K2 so that
// 157 row tiles still fill 256 CUs; all pieces are combined with fp64 atomics into zeroed G / ss.
#include "nrm_common.h"

#define SKM 128
#define SKN 32
#define SKK 32
#define SKP 34  // LDS pitch in doubles (272 B): 16-byte aligned rows, 16 rows of an operand read on distinct banks

typedef double d2_t __attribute__((ext_vector_type(2)));

struct SkinnySched {
	int nkt, tiles_dp, tiles_sk, units_per_wg, nwg;
};

template <typename T>
__device__ __forceinline__ void load4(const T* p, bool full, int64_t k, int64_t n, double (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* p, bool full, int64_t k, int64_t n, double (&v)[4]) {
	if (full) {
		float4 t = *reinterpret_cast<const float4*>(p + k);
		v[0] = t.x;
		v[1] = t.y;
		v[2] = t.z;
		v[3] = t.w;
	} else {
#pragma unroll
		for (int i = 0; i < 4; i++) v[i] = (k + i < n) ? (double)p[k + i] : 0.0;
	}
}
template <>
__device__ __forceinline__ void load4<double>(const double* p, bool full, int64_t k, int64_t n, double (&v)[4]) {
	if (full) {
		d2_t a = *reinterpret_cast<const d2_t*>(p + k), b = *reinterpret_cast<const d2_t*>(p + k + 2);
		v[0] = a[0];
		v[1] = a[1];
		v[2] = b[0];
		v[3] = b[1];
	} else {
#pragma unroll
		for (int i = 0; i < 4; i++) v[i] = (k + i < n) ? p[k + i] : 0.0;
	}
}

template <typename T>
__global__ void __launch_bounds__(256, 3) k_gram_skinny(const T* __restrict__ A, int64_t rows, int64_t n, int64_t lda,
														 const double* __restrict__ Z, int64_t ldz, double* __restrict__ G,
														 double* __restrict__ ss, SkinnySched s) {
	__shared__ __attribute__((aligned(16))) double lds[(SKM + SKN) * SKP];
	double* ldsA = lds;
	double* ldsZ = lds + SKM * SKP;
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int l15 = lane & 15, lg = lane >> 4;
	const int srow = tid >> 3, scol = (tid & 7) * 4;
	const int per_xcd = s.nwg >> 3;
	const int p = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
	int t_dp = p;
	int64_t u = (int64_t)p * s.units_per_wg;
	const int64_t total = (int64_t)s.tiles_sk * s.nkt;
	int64_t uend = u + s.units_per_wg;
	if (uend > total) uend = total;
	for (;;) {
		int t, k0, k1;
		if (t_dp < s.tiles_dp) {
			t = t_dp;
			k0 = 0;
			k1 = s.nkt;
			t_dp += s.nwg;
		} else if (u < uend) {
			const int ts = (int)(u / s.nkt);
			k0 = (int)(u - (int64_t)ts * s.nkt);
			int64_t k1l = k0 + (uend - u);
			k1 = k1l > s.nkt ? s.nkt : (int)k1l;
			t = s.tiles_dp + ts;
			u += k1 - k0;
		} else {
			break;
		}
		// ---- one piece: rows [t*128, +128), k-tiles [k0, k1) ----
		const T* arow[4];
		bool alive[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int64_t r = (int64_t)t * SKM + srow + 32 * j;
			alive[j] = r < rows;
			arow[j] = A + (alive[j] ? r : 0) * lda;
		}
		const double* zrow = Z + (int64_t)srow * ldz;
		double ra[4][4], rz[4], sq[4] = {0.0, 0.0, 0.0, 0.0};
		d4_t acc[2][2];
#pragma unroll
		for (int i = 0; i < 2; i++)
#pragma unroll
			for (int j = 0; j < 2; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
		{
			const int64_t k = (int64_t)k0 * SKK + scol;
			const bool full = k + 3 < n;
#pragma unroll
			for (int j = 0; j < 4; j++) load4<T>(arow[j], full, k, n, ra[j]);
			load4<double>(zrow, true, k, 0, rz);
		}
		for (int kt = k0; kt < k1; kt++) {
			__syncthreads();  // previous slab fully consumed
#pragma unroll
			for (int j = 0; j < 4; j++) {
				if (!alive[j]) ra[j][0] = ra[j][1] = ra[j][2] = ra[j][3] = 0.0;
#pragma unroll
				for (int i = 0; i < 4; i++) sq[j] = fma(ra[j][i], ra[j][i], sq[j]);
				double* d = &ldsA[(srow + 32 * j) * SKP + scol];
				*reinterpret_cast<d2_t*>(d) = (d2_t){ra[j][0], ra[j][1]};
				*reinterpret_cast<d2_t*>(d + 2) = (d2_t){ra[j][2], ra[j][3]};
			}
			{
				double* d = &ldsZ[srow * SKP + scol];
				*reinterpret_cast<d2_t*>(d) = (d2_t){rz[0], rz[1]};
				*reinterpret_cast<d2_t*>(d + 2) = (d2_t){rz[2], rz[3]};
			}
			__syncthreads();
			if (kt + 1 < k1) {  // prefetch the next slab while this one is contracted
				const int64_t k = (int64_t)(kt + 1) * SKK + scol;
				const bool full = k + 3 < n;
#pragma unroll
				for (int j = 0; j < 4; j++) load4<T>(arow[j], full, k, n, ra[j]);
				load4<double>(zrow, true, k, 0, rz);
			}
			const double* la = &ldsA[(wid * 32 + l15) * SKP + lg];
			const double* lz = &ldsZ[l15 * SKP + lg];
#pragma unroll
			for (int kk = 0; kk < SKK / 4; kk++) {
				const double a0 = la[kk * 4], a1 = la[16 * SKP + kk * 4];
				const double z0 = lz[kk * 4], z1 = lz[16 * SKP + kk * 4];
				acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, z0, acc[0][0], 0, 0, 0);
				acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, z1, acc[0][1], 0, 0, 0);
				acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, z0, acc[1][0], 0, 0, 0);
				acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, z1, acc[1][1], 0, 0, 0);
			}
		}
		// combine: G and ss start from zero (memset by the launcher)
		double* gbase = G + ((int64_t)t * SKM + wid * 32) * SKN;
#pragma unroll
		for (int i = 0; i < 2; i++)
#pragma unroll
			for (int j = 0; j < 2; j++)
#pragma unroll
				for (int q = 0; q < 4; q++) unsafeAtomicAdd(&gbase[(int64_t)(i * 16 + lg + 4 * q) * SKN + j * 16 + l15], acc[i][j][q]);
#pragma unroll
		for (int j = 0; j < 4; j++) {
			double v = sq[j];
			v += __shfl_xor(v, 1, 64);
			v += __shfl_xor(v, 2, 64);
			v += __shfl_xor(v, 4, 64);
			if ((tid & 7) == 0) unsafeAtomicAdd(&ss[(int64_t)t * SKM + srow + 32 * j], v);
		}
	}
}

static int g_num_cu_s = 0;

extern "C" int nrm_gram_skinny(const void* d_a, int a_dtype, int64_t rows, int64_t n, int64_t lda, const double* d_z, int64_t ldz,
							   int64_t k_pad, double* d_g, double* d_ss, int64_t rows_pad, void* stream) {
	NRM_REQUIRE(a_dtype == NRM_F32 || a_dtype == NRM_F64, "nrm_gram_skinny: bad dtype");
	NRM_REQUIRE(rows > 0 && n > 0 && lda >= n, "Incorrect dx/dy/dc size.");
	NRM_REQUIRE(k_pad >= n && k_pad % SKK == 0 && ldz >= k_pad && ldz % 2 == 0, "nrm_gram_skinny: Z must be padded to a multiple of %d cells", SKK);
	NRM_REQUIRE(rows_pad >= rows && rows_pad % SKM == 0, "nrm_gram_skinny: rows_pad must be a multiple of %d", SKM);
	NRM_REQUIRE(d_a && d_z && d_g && d_ss, "nrm_gram_skinny: null pointer");
	const int64_t al = 16 / (a_dtype == NRM_F64 ? 8 : 4);
	NRM_REQUIRE(lda % al == 0 && (uintptr_t)d_a % 16 == 0 && (uintptr_t)d_z % 16 == 0, "nrm_gram_skinny: rows must be 16-byte aligned");
	hipStream_t st = (hipStream_t)stream;
	if (g_num_cu_s == 0) {
		int dev = 0;
		NRM_HIP(hipGetDevice(&dev));
		NRM_HIP(hipDeviceGetAttribute(&g_num_cu_s, hipDeviceAttributeMultiprocessorCount, dev));
		if (g_num_cu_s <= 0) g_num_cu_s = 256;
	}
	NRM_HIP(hipMemsetAsync(d_g, 0, (size_t)rows_pad * SKN * sizeof(double), st));
	NRM_HIP(hipMemsetAsync(d_ss, 0, (size_t)rows_pad * sizeof(double), st));
	SkinnySched s;
	const int64_t tiles = rows_pad / SKM;
	s.nkt = (int)(k_pad / SKK);
	s.nwg = 3 * g_num_cu_s;
	s.nwg -= s.nwg % 8;
	const int64_t waves = tiles / s.nwg, rem = tiles - waves * s.nwg;
	int64_t sk = rem;
	if (rem > 0 && rem < s.nwg / 4 && waves >= 1) sk = rem + s.nwg;
	s.tiles_sk = (int)sk;
	s.tiles_dp = (int)(tiles - sk);
	s.units_per_wg = (int)((sk * s.nkt + s.nwg - 1) / s.nwg);
	if (a_dtype == NRM_F64)
		hipLaunchKernelGGL(k_gram_skinny<double>, dim3((unsigned)s.nwg), dim3(256), 0, st, (const double*)d_a, rows, n, lda, d_z, ldz, d_g, d_ss, s);
	else
		hipLaunchKernelGGL(k_gram_skinny<float>, dim3((unsigned)s.nwg), dim3(256), 0, st, (const float*)d_a, rows, n, lda, d_z, ldz, d_g, d_ss, s);
	return nrm_check_launch("k_gram_skinny");
}
