// K2: dot[i,j] = sum_k A[i,k] B[j,k] on the fp64 matrix cores of gfx950 (v_mfma_f64_16x16x4_f64).
// Replaces np.matmul(dy1, dx1.T) at association.py:234 for the whole problem in one launch.
//
// Both operands are K-contiguous (gene-major, cells contiguous: association.py:163-170), so this is an
// "NT" GEMM whose global loads of A and B are both coalesced along K.
//
// Geometry (MFMA-bound; 2*n_cell flop per pair):
//   workgroup tile 128 x 128, 256 threads = 4 waves as 2(M) x 2(N); each wave owns 64 x 64 = 4 x 4 MFMA
//   tiles (64 fp64 accumulators per lane, kept in VGPRs: with AGPR accumulators this instruction issues
//   at half rate on MI355X, tools/mfma_peak.hip).  K is consumed in slabs of GK = 16 cells staged through
//   a double-buffered LDS image [row][16 cells], rows unpadded (128 B = one line), whose 16-byte chunks are
//   XOR-swizzled with row & 7 so that the 8 rows served per clock of a 16-byte operand read cover all banks.
//   MFMA operand maps (f64 16x16x4): lane l supplies A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15];
//   it receives D[row = (l >> 4) + 4 q][col = l & 15] in accumulator element q.
//   Staging is global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write; each wave
//   instruction fills 8 rows x 128 B linearly, the swizzle is applied to the source chunk a lane fetches): the next
//   slab's loads are issued before the MFMA block of the current slab into the other buffer, one barrier per slab
//   (it also drains vmcnt); two workgroups per CU (64 KB LDS each) cover each other's barrier bubbles.
//   Measured on C2 (tools/k2_time.py, experiments with parts of the loop compiled out): MFMA + LDS reads alone
//   72.5 TF executed (= the instruction's ceiling on this chip); register-staged loads cost 5 %, the ds_writes 3 %,
//   the barrier 7 % -> 63.9 TF; direct-to-LDS staging removes the first two: 68.3 TF executed.
//
// Scheduling (persistent, "data-parallel + stream-K"): the launch is 2 workgroups per CU.  Whole waves of
// tiles are processed tile-per-workgroup with all workgroups in K-lockstep (operand slabs shared through
// L2); the tiles of the last partial wave are cut along K into equal unit ranges so every workgroup
// finishes together (820 tiles on 512 slots would otherwise idle 20 % of the chip).  Tile pieces that do
// not cover the whole K range are written to workspace slabs and summed per tile in a fixed order by a small
// fix-up kernel (deterministic: no atomics).
// Symmetric (coex) launches only enumerate tiles on or above the block diagonal (association.py:893-894).
#include "nrm_gram_sched.h"

#define GK 16

// One tile piece: k-tiles [kt0, kt1) of tile (ti, tj).  slab != null -> partial piece, stored to its workspace slab.
__device__ __forceinline__ void gram_piece(const double* __restrict__ A, const double* __restrict__ B, int64_t lda, int64_t ldb,
										   double* __restrict__ C, int64_t ldc, int ti, int tj, int kt0, int kt1, double* __restrict__ slab,
										   const unsigned need /* bit i*4+j: this wave's 16x16 sub-block (i,j) is wanted */,
										   double* lds /* [2][2][GM*GK] */) {
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int wm = wid >> 1, wn = wid & 1;
	const int l15 = lane & 15, lg = lane >> 4;
	const int nk = kt1 - kt0;
	// Staging: global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write).  One wave instruction
	// writes 1 KB of LDS linearly = 8 rows x 128 B of a slab (16 cells of a row = one 128-byte line, 8 chunks of 16 B);
	// rows are unpadded, so the 16-byte chunks of row r are stored XOR-swizzled with r & 7 -- applied to the SOURCE
	// chunk each lane fetches and again to the chunk index of every operand read (same involution on both sides) -- which
	// spreads the 16 rows of an MFMA operand read over all banks.  Wave w stages rows [32 w, 32 w + 32) of A and of B.
	typedef const __attribute__((address_space(1))) void* gptr_t;
	typedef __attribute__((address_space(3))) void* lptr_t;
	const int jr = lane >> 3, jc = lane & 7;  // row within the instruction's 8 rows, physical chunk
	const double* ga[4];
	const double* gb[4];
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const int r = wid * 32 + q * 8 + jr;
		const int c = jc ^ (r & 7);
		ga[q] = A + ((int64_t)ti * GM + r) * lda + (int64_t)kt0 * GK + c * 2;
		gb[q] = B + ((int64_t)tj * GN + r) * ldb + (int64_t)kt0 * GK + c * 2;
	}
	double* const ldsA0 = lds;
	double* const ldsB0 = lds + GM * GK;
	double* const ldsA1 = lds + 2 * GM * GK;
	double* const ldsB1 = lds + 3 * GM * GK;
	const int wrow = wid * 32 * GK;  // this wave's staging rows (doubles)
#define GRAM_STAGE(dstA, dstB, koff)                                                                                              \
	_Pragma("unroll") for (int q = 0; q < 4; q++) {                                                                               \
		__builtin_amdgcn_global_load_lds((gptr_t)(ga[q] + (koff)), (lptr_t)((dstA) + wrow + q * 8 * GK), 16, 0, 0);              \
		__builtin_amdgcn_global_load_lds((gptr_t)(gb[q] + (koff)), (lptr_t)((dstB) + wrow + q * 8 * GK), 16, 0, 0);              \
	}
	GRAM_STAGE(ldsA0, ldsB0, 0)

	d4_t acc[4][4];
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int j = 0; j < 4; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};

	// operand reads: 16 bytes (two cells) per lane and read.  Lane group lg reads chunks lg and 4 + lg of its row; the
	// first cells of the four lane groups feed one MFMA, the second cells the next -- a permutation of the 16 cells of
	// the slab that is the same for A and B, so the contraction is unchanged.  Eight consecutive lanes (rows r..r+7, one
	// chunk each, swizzled with r & 7) cover all 32 banks: conflict-free ds_read_b128.
	const int c0 = (lg ^ (l15 & 7)) * 2, c1 = ((4 + lg) ^ (l15 & 7)) * 2;
	const int aoff = (wm * 64 + l15) * GK;
	const int boff = (wn * 64 + l15) * GK;
	__syncthreads();

	for (int kt = 0; kt < nk; kt++) {
		const int cur = kt & 1;
		if (kt + 1 < nk) {
			const int64_t ko = (int64_t)(kt + 1) * GK;
			if (cur) {
				GRAM_STAGE(ldsA0, ldsB0, ko)
			} else {
				GRAM_STAGE(ldsA1, ldsB1, ko)
			}
		}
		const double* la = (cur ? ldsA1 : ldsA0) + aoff;
		const double* lb = (cur ? ldsB1 : ldsB0) + boff;
		// (predicating individual MFMAs on `need` was measured: the 16 scalar branches per k-step cost 3 % on full tiles,
		//  more than the 5 % of padded / mirrored sub-blocks they save on 80 of 820 tiles -- so the loop stays branch-free)
#pragma unroll
		for (int h = 0; h < 2; h++) {
			d2_t fa[4], fb[4];
#pragma unroll
			for (int i = 0; i < 4; i++) {
				fa[i] = *reinterpret_cast<const d2_t*>(la + i * 16 * GK + (h ? c1 : c0));
				fb[i] = *reinterpret_cast<const d2_t*>(lb + i * 16 * GK + (h ? c1 : c0));
			}
#pragma unroll
			for (int e = 0; e < 2; e++)
#pragma unroll
				for (int i = 0; i < 4; i++)
#pragma unroll
					for (int j = 0; j < 4; j++)
						acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
		}
		__syncthreads();  // drains the slab in flight (vmcnt) and fences the buffer just read
	}
#undef GRAM_STAGE

	// epilogue: lane l holds D[row = lg + 4 q][col = l15] of each 16x16 tile.  A piece that covers the whole K range goes
	// straight to C; a partial piece goes to its own 128x128 slab of the workspace and k_gram_fixup adds the slabs of a
	// tile in a fixed order (no atomics: results are bitwise reproducible from run to run).
	double* cbase;
	int64_t pitch;
	if (slab) {
		cbase = slab + (wm * 64) * GN + wn * 64;
		pitch = GN;
	} else {
		cbase = C + ((int64_t)ti * GM + wm * 64) * ldc + (int64_t)tj * GN + wn * 64;
		pitch = ldc;
	}
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int q = 0; q < 4; q++)
				if (slab || (need & (1u << (i * 4 + j)))) cbase[(int64_t)(i * 16 + lg + 4 * q) * pitch + j * 16 + l15] = acc[i][j][q];
}

__global__ void __launch_bounds__(256, 2) k_gram_f64(const double* __restrict__ A, const double* __restrict__ B, int64_t lda,
													  int64_t ldb, double* __restrict__ C, int64_t ldc, int symmetric, GramSched s) {
	__shared__ __attribute__((aligned(16))) double lds[2 * 2 * GM * GK];
	gram_for_each_piece(s, [&](int t, int k0, int k1, double* slab) {
		int ti, tj;
		gram_tile_coords(s.tile0 + t, symmetric, s.ntm, s.ntn, ti, tj);
		// which of this wave's 4x4 sub-blocks are wanted: rows/columns inside the matrix and, on diagonal tiles of a
		// symmetric problem, not strictly below the diagonal (K3 only reads dot[min(i,j)][max(i,j)])
		unsigned need = 0;
		{
			const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;
			const int r0 = ti * GM + wm * 64, c0 = tj * GN + wn * 64;
			const bool diag = symmetric && ti == tj;
#pragma unroll
			for (int i = 0; i < 4; i++)
#pragma unroll
				for (int j = 0; j < 4; j++)
					if (r0 + i * 16 < s.m_rows && c0 + j * 16 < s.n_rows && (!diag || c0 + j * 16 >= r0 + i * 16)) need |= 1u << (i * 4 + j);
			need = __builtin_amdgcn_readfirstlane(need);
		}
		gram_piece(A, B, lda, ldb, C, ldc, ti, tj, k0, k1, slab, need, lds);
	});
}

static int g_num_cu = 0;

extern "C" int64_t nrm_gram_workspace_bytes(void) {
	if (g_num_cu == 0) {
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&g_num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || g_num_cu <= 0)
			g_num_cu = 256;
	}
	// (the fp64 kernel runs two workgroups per CU, the integer engine one)
	return nrm_host_gram_workspace_doubles(2 * g_num_cu) * (int64_t)sizeof(double);
}

extern "C" int nrm_gram_f64_band(const double* d_a, const double* d_b, int64_t m_pad, int64_t n_pad, int64_t k_pad, int64_t lda,
								 int64_t ldb, double* d_dot, int64_t ldd, int symmetric, int64_t m_rows, int64_t n_rows, int64_t row0,
								 int64_t row1, void* d_work, void* stream);

extern "C" int nrm_gram_f64(const double* d_a, const double* d_b, int64_t m_pad, int64_t n_pad, int64_t k_pad, int64_t lda,
							int64_t ldb, double* d_dot, int64_t ldd, int symmetric, int64_t m_rows, int64_t n_rows, void* d_work,
							void* stream) {
	return nrm_gram_f64_band(d_a, d_b, m_pad, n_pad, k_pad, lda, ldb, d_dot, ldd, symmetric, m_rows, n_rows, 0, m_pad, d_work, stream);
}

extern "C" int nrm_gram_f64_band(const double* d_a, const double* d_b, int64_t m_pad, int64_t n_pad, int64_t k_pad, int64_t lda,
								 int64_t ldb, double* d_dot, int64_t ldd, int symmetric, int64_t m_rows, int64_t n_rows, int64_t row0,
								 int64_t row1, void* d_work, void* stream) {
	NRM_REQUIRE(m_pad >= 0 && n_pad >= 0 && k_pad > 0, "nrm_gram_f64: bad sizes");
	NRM_REQUIRE(m_pad % GM == 0 && n_pad % GN == 0 && k_pad % GK == 0, "nrm_gram_f64: sizes must be padded to %d/%d/%d", GM, GN, GK);
	NRM_REQUIRE(lda >= k_pad && ldb >= k_pad && ldd >= n_pad, "nrm_gram_f64: pitch too small");
	NRM_REQUIRE(lda % 2 == 0 && ldb % 2 == 0 && ldd % 2 == 0, "nrm_gram_f64: pitches must be even (16-byte rows)");
	NRM_REQUIRE(!symmetric || m_pad == n_pad, "nrm_gram_f64: symmetric needs square output");
	NRM_REQUIRE(row0 >= 0 && row0 <= row1 && row1 <= m_pad && row0 % (GSB * GM) == 0 && (row1 % (GSB * GM) == 0 || row1 == m_pad),
				"nrm_gram_f64_band: rows [row0, row1) must be cut at multiples of %d", GSB * GM);
	if (m_pad == 0 || n_pad == 0 || row0 == row1) return NRM_OK;
	NRM_REQUIRE(d_a && d_b && d_dot, "nrm_gram_f64: null pointer");
	NRM_REQUIRE(((uintptr_t)d_a % 16 == 0) && ((uintptr_t)d_b % 16 == 0), "nrm_gram_f64: operands must be 16-byte aligned");
	if (g_num_cu == 0) {
		int dev = 0;
		NRM_HIP(hipGetDevice(&dev));
		NRM_HIP(hipDeviceGetAttribute(&g_num_cu, hipDeviceAttributeMultiprocessorCount, dev));
		if (g_num_cu <= 0) g_num_cu = 256;
	}
	NRM_REQUIRE(d_work != nullptr, "nrm_gram_f64: workspace of nrm_gram_workspace_bytes() bytes required");
	GramSched s;
	NRM_TRY_RC(gram_plan(s, m_pad, n_pad, k_pad / GK, symmetric, m_rows, n_rows, row0, row1, 2 * g_num_cu, (double*)d_work));
	hipLaunchKernelGGL(k_gram_f64, dim3((unsigned)s.nwg), dim3(256), 0, (hipStream_t)stream, d_a, d_b, lda, ldb, d_dot, ldd, symmetric, s);
	if (s.tiles_al + s.tiles_sk > 0)
		hipLaunchKernelGGL(k_gram_fixup<0>, dim3((unsigned)(s.tiles_al + s.tiles_sk), GM / GFIX_ROWS), dim3(256), 0, (hipStream_t)stream, d_dot, ldd, symmetric, s);
	return nrm_check_launch("k_gram_f64");
}
