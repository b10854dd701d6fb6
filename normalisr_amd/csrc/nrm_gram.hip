// K2: dot[i,j] = sum_k A[i,k] B[j,k] on the fp64 matrix cores of gfx950 (v_mfma_f64_16x16x4_f64).
// Replaces np.matmul(dy1, dx1.T) at association.py:234 for the whole problem in one launch.
//
// Both operands are K-contiguous (gene-major, cells contiguous: association.py:163-170), so this is an
// "NT" GEMM whose global loads of A and B are both coalesced along K.
//
// Geometry (MFMA-bound; 2*n_cell flop per pair):
//   workgroup tile 128 x 128, 256 threads = 4 waves as 2(M) x 2(N); each wave owns 64 x 64 = 4 x 4 MFMA
//   tiles (64 fp64 accumulators per lane).  K is consumed in slabs of GK = 16 cells staged through a
//   double-buffered LDS image [row][k] with an 18-element pitch: with 144-byte rows the 16 rows a
//   ds_read_b64 wave-instruction touches per half-wave land on 16 distinct 4-bank groups (conflict-free),
//   and every 16-byte staging store stays 16-byte aligned.
//   MFMA operand maps (f64 16x16x4): lane l supplies A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15];
//   it receives D[row = (l >> 4) + 4 q][col = l & 15] in accumulator element q.
//   Global -> register -> LDS staging: next slab's loads are issued before the MFMA block of the current
//   slab and stored to the other LDS buffer after it (one barrier per slab); two workgroups per CU
//   (74 KB LDS each) cover each other's barrier bubbles.
//   Symmetric (coex) launches only enumerate tiles on or above the block diagonal (association.py:893-894).
#include "nrm_common.h"

#define GM 128
#define GN 128
#define GK 16
#define GP 18  // LDS row pitch in doubles (144 B)

typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void gram_tile_coords(int symmetric, int ntn, int& ti, int& tj) {
	if (!symmetric) {
		ti = blockIdx.y;
		tj = blockIdx.x;
		return;
	}
	// linear index over the upper triangle, row by row: row i holds tiles (i, i..ntn-1)
	int b = blockIdx.x;
	int i = 0, len = ntn;
	while (b >= len) {
		b -= len;
		i++;
		len--;
	}
	ti = i;
	tj = i + b;
}

__global__ void __launch_bounds__(256, 2) k_gram_f64(const double* __restrict__ A, const double* __restrict__ B, int64_t lda,
													  int64_t ldb, int nk, double* __restrict__ C, int64_t ldc, int symmetric,
													  int ntn) {
	__shared__ __attribute__((aligned(16))) double lds[2][2][GM * GP];  // [buffer][A|B][row*GP + k]
	int ti, tj;
	gram_tile_coords(symmetric, ntn, ti, tj);
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int wm = wid >> 1, wn = wid & 1;
	const int l15 = lane & 15, lg = lane >> 4;

	// staging map: 8 consecutive lanes cover one 128-byte row slab; 4 passes of 32 rows
	const int srow = tid >> 3, scol = (tid & 7) * 2;
	const double* ga = A + ((int64_t)ti * GM + srow) * lda + scol;
	const double* gb = B + ((int64_t)tj * GN + srow) * ldb + scol;
	const int soff = srow * GP + scol;

	d2_t ra[4], rb[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		ra[j] = *reinterpret_cast<const d2_t*>(ga + (int64_t)j * 32 * lda);
		rb[j] = *reinterpret_cast<const d2_t*>(gb + (int64_t)j * 32 * ldb);
	}
#pragma unroll
	for (int j = 0; j < 4; j++) {
		*reinterpret_cast<d2_t*>(&lds[0][0][soff + j * 32 * GP]) = ra[j];
		*reinterpret_cast<d2_t*>(&lds[0][1][soff + j * 32 * GP]) = rb[j];
	}
	__syncthreads();

	d4_t acc[4][4];
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int j = 0; j < 4; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};

	const int aoff = (wm * 64 + l15) * GP + lg;
	const int boff = (wn * 64 + l15) * GP + lg;

	for (int kt = 0; kt < nk; kt++) {
		const int cur = kt & 1;
		const bool more = kt + 1 < nk;
		if (more) {
			const int64_t ko = (int64_t)(kt + 1) * GK;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				ra[j] = *reinterpret_cast<const d2_t*>(ga + (int64_t)j * 32 * lda + ko);
				rb[j] = *reinterpret_cast<const d2_t*>(gb + (int64_t)j * 32 * ldb + ko);
			}
		}
		const double* la = &lds[cur][0][aoff];
		const double* lb = &lds[cur][1][boff];
#pragma unroll
		for (int kk = 0; kk < GK / 4; kk++) {
			double fa[4], fb[4];
#pragma unroll
			for (int i = 0; i < 4; i++) {
				fa[i] = la[i * 16 * GP + kk * 4];
				fb[i] = lb[i * 16 * GP + kk * 4];
			}
#pragma unroll
			for (int i = 0; i < 4; i++)
#pragma unroll
				for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
		}
		if (more) {
#pragma unroll
			for (int j = 0; j < 4; j++) {
				*reinterpret_cast<d2_t*>(&lds[cur ^ 1][0][soff + j * 32 * GP]) = ra[j];
				*reinterpret_cast<d2_t*>(&lds[cur ^ 1][1][soff + j * 32 * GP]) = rb[j];
			}
		}
		__syncthreads();
	}

	// epilogue: lane l holds D[row = lg + 4 q][col = l15] of each 16x16 tile
	double* cbase = C + ((int64_t)ti * GM + wm * 64) * ldc + (int64_t)tj * GN + wn * 64;
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int q = 0; q < 4; q++) cbase[(int64_t)(i * 16 + lg + 4 * q) * ldc + j * 16 + l15] = acc[i][j][q];
}

extern "C" int nrm_gram_f64(const double* d_a, const double* d_b, int64_t m_pad, int64_t n_pad, int64_t k_pad, int64_t lda,
							int64_t ldb, double* d_dot, int64_t ldd, int symmetric, void* stream) {
	NRM_REQUIRE(m_pad >= 0 && n_pad >= 0 && k_pad > 0, "nrm_gram_f64: bad sizes");
	NRM_REQUIRE(m_pad % GM == 0 && n_pad % GN == 0 && k_pad % GK == 0, "nrm_gram_f64: sizes must be padded to %d/%d/%d", GM, GN, GK);
	NRM_REQUIRE(lda >= k_pad && ldb >= k_pad && ldd >= n_pad, "nrm_gram_f64: pitch too small");
	NRM_REQUIRE(lda % 2 == 0 && ldb % 2 == 0, "nrm_gram_f64: operand pitches must be even (16-byte rows)");
	NRM_REQUIRE(!symmetric || m_pad == n_pad, "nrm_gram_f64: symmetric needs square output");
	if (m_pad == 0 || n_pad == 0) return NRM_OK;
	NRM_REQUIRE(d_a && d_b && d_dot, "nrm_gram_f64: null pointer");
	NRM_REQUIRE(((uintptr_t)d_a % 16 == 0) && ((uintptr_t)d_b % 16 == 0), "nrm_gram_f64: operands must be 16-byte aligned");
	int ntm = (int)(m_pad / GM), ntn = (int)(n_pad / GN);
	dim3 grid;
	if (symmetric)
		grid = dim3((unsigned)((int64_t)ntn * (ntn + 1) / 2));
	else
		grid = dim3((unsigned)ntn, (unsigned)ntm);
	hipLaunchKernelGGL(k_gram_f64, grid, dim3(256), 0, (hipStream_t)stream, d_a, d_b, lda, ldb, (int)(k_pad / GK), d_dot, ldd,
					   symmetric, ntn);
	return nrm_check_launch("k_gram_f64");
}
