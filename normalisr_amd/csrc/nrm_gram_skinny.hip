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
// Algorithmic HBM bytes: itemsize * n per expression row (+ Z from L2).
//
// Geometry: a wave owns 64 expression rows (4 MFMA row tiles) x NT*16 Z rows.  The Y operand goes from
// global memory STRAIGHT into the MFMA A-operand layout -- no LDS round trip, no barrier: for a slab of 16
// cells lane (r = l & 15, g = l >> 4) loads the 4 consecutive cells 4g..4g+3 of row r (one 16-byte load for
// fp32), and MFMA step s = 0..3 consumes cell 4g+s; a dot product does not care about the order of its
// terms as long as the Z operand uses the same permutation (it does: lane (z, g) reads cells 4g..4g+3 of Z
// row z from LDS).  Slabs are prefetched 4 (fp32) / 2 (fp64) deep in registers to cover HBM latency (accumulators
// forced into VGPRs: -amdgpu-mfma-vgpr-form, AGPR accumulators halve the fp64 MFMA rate; 240-256 VGPRs, so two
// workgroups share a CU).  Measured by compiling parts out (tools/k2s_time.py, C3 shape, 21 Z rows): everything but
// MFMA + LDS reads removed 1.77 ms (72.5 TF, the instruction ceiling for 32 padded Z rows); the expression-row loads
// cost 0.6 ms, re-staging Z 0.5 ms, the chunk barrier 0.15 ms -> 2.73 ms.  Z (tiny, shared
// by every wave) is staged through LDS in chunks of 128 cells, double buffered, one barrier per chunk.
// Persistent DP + stream-K schedule as in K2 so that 79 row tiles still fill 256 CUs; partial pieces go to
// workspace slabs and are summed in a fixed order (k_skinny_fixup): bitwise reproducible, no atomics.
#include "nrm_common.h"

#define SKM 256      // rows per workgroup tile (4 waves x 64)
#define SKN 32       // columns of G
#define SKC 128      // cells per Z chunk in LDS
#define SKP (SKC + 2)  // LDS pitch in doubles: 16-byte aligned, rows 16 bytes apart modulo the bank row

typedef double d2_t __attribute__((ext_vector_type(2)));

struct SkinnySched {
	int nkt, tiles_dp, tiles_sk, units_per_wg, nwg;
	double* work;  // per partial piece: a (256 x 32) slab of G followed by 256 sums of squares; two pieces per workgroup
};
#define SK_SLAB (SKM * SKN + SKM)

template <typename T>
struct Slab;  // 4 consecutive cells of one row, as loaded (rows are readable and zero up to a multiple of 16 cells)
template <>
struct Slab<float> {
	float4 v;
	__device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
	__device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
	__device__ __forceinline__ double get(int s) const { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }
};
template <>
struct Slab<double> {
	d2_t a, b;
	__device__ __forceinline__ void load(const double* p) {
		a = *reinterpret_cast<const d2_t*>(p);
		b = *reinterpret_cast<const d2_t*>(p + 2);
	}
	__device__ __forceinline__ void zero() {
		a = (d2_t){0.0, 0.0};
		b = a;
	}
	__device__ __forceinline__ double get(int s) const { return s == 0 ? a[0] : s == 1 ? a[1] : s == 2 ? b[0] : b[1]; }
};

// NT = number of 16-row Z tiles contracted with v_mfma_f64_16x16x4 (1 when nx + nc <= 16: half the MFMA work);
// NQ = number of further 4-row Z groups contracted with v_mfma_f64_4x4x4 (four independent 4x4x4 blocks per
// instruction, 16 instead of 64 cycles): 17-24 Z rows cost 64 + NQ * 16 cycles per step instead of 128.  Its A operand
// map is the 16x16x4 one (lane 16 k + r holds row r, cell k; block = r >> 2), so the expression registers feed both;
// its B operand lane 16 k + 4 b + j holds Z[4 g + j][k] for every block b (an LDS broadcast); it returns
// D[row 4 b + i][z 4 g + j] in lane 16 i + 4 b + j (tools/mfma444_probe.hip).
template <typename T, int NT, int NQ>
__global__ void __launch_bounds__(256, 1) k_gram_skinny(const T* __restrict__ A, int64_t rows, int64_t n16, int64_t lda,
														 const double* __restrict__ Z, int64_t ldz, double* __restrict__ G,
														 double* __restrict__ ss, SkinnySched s) {
	constexpr int ZR = NT * 16 + NQ * 4;  // Z rows held in LDS
	__shared__ __attribute__((aligned(16))) double lds[2][ZR * SKP];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int l15 = lane & 15, lg = lane >> 4;
	const int per_xcd = s.nwg >> 3;
	const int p = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
	int t_dp = p;
	int64_t u = (int64_t)p * s.units_per_wg;
	const int64_t total = (int64_t)s.tiles_sk * s.nkt;
	int64_t uend = u + s.units_per_wg;
	if (uend > total) uend = total;
	int sk_piece = 0;
	for (;;) {
		int t, c0, c1;  // tile, chunk range [c0, c1) in units of SKC cells
		double* slab = nullptr;
		if (t_dp < s.tiles_dp) {
			t = t_dp;
			c0 = 0;
			c1 = s.nkt;
			t_dp += s.nwg;
		} else if (u < uend) {
			const int ts = (int)(u / s.nkt);
			c0 = (int)(u - (int64_t)ts * s.nkt);
			int64_t c1l = c0 + (uend - u);
			c1 = c1l > s.nkt ? s.nkt : (int)c1l;
			t = s.tiles_dp + ts;
			u += c1 - c0;
			if (!(c0 == 0 && c1 == s.nkt)) slab = s.work + ((int64_t)2 * p + sk_piece) * SK_SLAB;
			sk_piece++;
		} else {
			break;
		}
		// ---- one piece: rows [t*256 + wid*64, +64) of this wave, cells [c0*128, c1*128) ----
		// rows past the end are clamped to row 0: their products land in padding rows of G / ss that nobody reads
		const T* arow[4];
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int64_t r = (int64_t)t * SKM + wid * 64 + i * 16 + l15;
			arow[i] = A + (r < rows ? r : 0) * lda + 4 * lg;
		}
		d4_t acc[4][NT];
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int j = 0; j < NT; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
		double accq[4][NQ ? NQ : 1];
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int g = 0; g < (NQ ? NQ : 1); g++) accq[i][g] = 0.0;
		double sq[4] = {0.0, 0.0, 0.0, 0.0};
		const int64_t kbeg = (int64_t)c0 * SKC;
		constexpr int DEPTH = sizeof(T) == 4 ? 4 : 2;  // slabs (16 cells) in flight per wave: 16 KB per wave, 64 KB per CU
		Slab<T> pre[DEPTH][4];
#pragma unroll
		for (int d = 0; d < DEPTH; d++) {
			const int64_t k = kbeg + (int64_t)d * 16;
			if (k < n16) {
#pragma unroll
				for (int i = 0; i < 4; i++) pre[d][i].load(arow[i] + k);
			} else {
#pragma unroll
				for (int i = 0; i < 4; i++) pre[d][i].zero();
			}
		}
		// Z chunk: NT*16 rows x 128 cells; thread -> row tid/8 (+16 per pass), 16 consecutive cells: one base address, immediates
		const double* zsrc = Z + (int64_t)(tid >> 3) * ldz + (tid & 7) * 16;
		const int zdst = (tid >> 3) * SKP + (tid & 7) * 16;
		auto stage_z = [&](int buf, int64_t k0) {
			if ((tid >> 3) < ZR) {  // only the Z rows in use exist in LDS
				const double* src = zsrc + k0;
				double* dst = &lds[buf][zdst];
#pragma unroll
				for (int q = 0; q < 8; q++) *reinterpret_cast<d2_t*>(dst + 2 * q) = *reinterpret_cast<const d2_t*>(src + 2 * q);
			}
		};
		__syncthreads();  // previous piece done with LDS
		stage_z(0, kbeg);
		__syncthreads();
		for (int c = c0; c < c1; c++) {
			const int buf = (c - c0) & 1;
			if (c + 1 < c1) stage_z(buf ^ 1, (int64_t)(c + 1) * SKC);
			const double* zl = &lds[buf][l15 * SKP + 4 * lg];
			const double* zql = &lds[buf][(NT * 16 + (l15 & 3)) * SKP + 4 * lg];
			const int64_t kc = (int64_t)c * SKC;
#pragma unroll 1
			for (int h = 0; h < SKC / 16 / DEPTH; h++) {
#pragma unroll
				for (int q = 0; q < DEPTH; q++) {
					const int sl = h * DEPTH + q;
					__builtin_amdgcn_sched_barrier(0);  // keep each slab's LDS reads next to its MFMAs (register budget)
					Slab<T> cur[4];
#pragma unroll
					for (int i = 0; i < 4; i++) cur[i] = pre[q][i];
					// refill this slot with the slab DEPTH ahead (still inside this piece and inside the padded rows)
					const int64_t kn = kc + (int64_t)(sl + DEPTH) * 16;
					if (kn < n16 && kn < (int64_t)c1 * SKC) {
#pragma unroll
						for (int i = 0; i < 4; i++) pre[q][i].load(arow[i] + kn);
					} else {
#pragma unroll
						for (int i = 0; i < 4; i++) pre[q][i].zero();
					}
					double zf[NT][4];
#pragma unroll
					for (int j = 0; j < NT; j++) {
						const d2_t z01 = *reinterpret_cast<const d2_t*>(zl + j * 16 * SKP + sl * 16);
						const d2_t z23 = *reinterpret_cast<const d2_t*>(zl + j * 16 * SKP + sl * 16 + 2);
						zf[j][0] = z01[0];
						zf[j][1] = z01[1];
						zf[j][2] = z23[0];
						zf[j][3] = z23[1];
					}
					double zq[NQ ? NQ : 1][4];
#pragma unroll
					for (int g = 0; g < NQ; g++) {
						const d2_t z01 = *reinterpret_cast<const d2_t*>(zql + g * 4 * SKP + sl * 16);
						const d2_t z23 = *reinterpret_cast<const d2_t*>(zql + g * 4 * SKP + sl * 16 + 2);
						zq[g][0] = z01[0];
						zq[g][1] = z01[1];
						zq[g][2] = z23[0];
						zq[g][3] = z23[1];
					}
#pragma unroll
					for (int st = 0; st < 4; st++)
#pragma unroll
						for (int i = 0; i < 4; i++) {
							const double a = cur[i].get(st);
							sq[i] = fma(a, a, sq[i]);
#pragma unroll
							for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, zf[j][st], acc[i][j], 0, 0, 0);
#pragma unroll
							for (int g = 0; g < NQ; g++) accq[i][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, zq[g][st], accq[i][g], 0, 0, 0);
						}
				}
			}
			__syncthreads();  // next chunk staged, this buffer free
		}
		// a piece covering all cells writes G / ss directly; partial pieces write their slab, summed in fixed order by k_skinny_fixup
		double* gbase = slab ? slab + (wid * 64) * SKN : G + ((int64_t)t * SKM + wid * 64) * SKN;
		double* sbase = slab ? slab + SKM * SKN + wid * 64 : ss + (int64_t)t * SKM + wid * 64;
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int j = 0; j < NT; j++)
#pragma unroll
				for (int q = 0; q < 4; q++) gbase[(int64_t)(i * 16 + lg + 4 * q) * SKN + j * 16 + l15] = acc[i][j][q];
		if (NT < 2) {  // columns not covered by a 16-row tile: the 4-row groups, zeros beyond them
			const int yrow = 4 * ((lane >> 2) & 3) + lg, zc = lane & 3;
#pragma unroll
			for (int i = 0; i < 4; i++)
#pragma unroll
				for (int g = 0; g < 4; g++) {
					double v = 0.0;
#pragma unroll
					for (int gg = 0; gg < NQ; gg++)
						if (gg == g) v = accq[i][gg];
					gbase[(int64_t)(i * 16 + yrow) * SKN + 16 + 4 * g + zc] = v;
				}
		}
#pragma unroll
		for (int i = 0; i < 4; i++) {
			double v = sq[i];
			v += __shfl_xor(v, 16, 64);
			v += __shfl_xor(v, 32, 64);
			if (lg == 0) sbase[i * 16 + l15] = v;
		}
	}
}

// Sum the slabs of every split row tile in workgroup order (deterministic) into G and ss.  blockIdx.y selects a
// 256-element chunk of the slab so that enough loads are in flight.
__global__ void __launch_bounds__(256) k_skinny_fixup(double* __restrict__ G, double* __restrict__ ss, SkinnySched s) {
	const int ts = blockIdx.x;
	const int64_t u0 = (int64_t)ts * s.nkt, u1 = u0 + s.nkt;
	const int first = (int)(u0 / s.units_per_wg);
	int last = (int)((u1 - 1) / s.units_per_wg);
	if (last > s.nwg - 1) last = s.nwg - 1;
	if (first == last && (int64_t)first * s.units_per_wg <= u0 && (int64_t)(first + 1) * s.units_per_wg >= u1) return;  // written whole
	const int64_t t = s.tiles_dp + ts;
	const int e = (blockIdx.y * 256 + threadIdx.x) * 2;  // two consecutive elements per thread (16-byte loads)
	if (e >= SK_SLAB) return;
	// only the first contributing workgroup can have started in the previous tile (then this tile holds its second piece)
	const int first_local = ((int64_t)first * s.units_per_wg / s.nkt) == ts ? 0 : 1;
	const double* src = s.work + ((int64_t)2 * first + first_local) * SK_SLAB + e;
	d2_t acc = *reinterpret_cast<const d2_t*>(src);
	src += (int64_t)(2 - first_local) * SK_SLAB;  // first piece of workgroup first + 1
	for (int p = first + 1; p <= last; p++, src += 2 * SK_SLAB) acc += *reinterpret_cast<const d2_t*>(src);
	if (e < SKM * SKN)
		*reinterpret_cast<d2_t*>(G + t * SKM * SKN + e) = acc;
	else
		*reinterpret_cast<d2_t*>(ss + t * SKM + (e - SKM * SKN)) = acc;
}

static int g_num_cu_s = 0;

extern "C" int64_t nrm_gram_skinny_workspace_bytes(void) {
	if (g_num_cu_s == 0) {
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&g_num_cu_s, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || g_num_cu_s <= 0)
			g_num_cu_s = 256;
	}
	return (int64_t)2 * (2 * g_num_cu_s) * SK_SLAB * (int64_t)sizeof(double);
}

extern "C" int nrm_gram_skinny(const void* d_a, int a_dtype, int64_t rows, int64_t n, int64_t lda, const double* d_z, int64_t ldz,
							   int64_t k_pad, double* d_g, double* d_ss, int64_t rows_pad, int64_t nz, void* d_work, void* stream) {
	NRM_REQUIRE(a_dtype == NRM_F32 || a_dtype == NRM_F64, "nrm_gram_skinny: bad dtype");
	NRM_REQUIRE(rows > 0 && n > 0 && lda >= n, "Incorrect dx/dy/dc size.");
	const int64_t n16 = (n + 15) / 16 * 16;
	NRM_REQUIRE(lda >= n16, "nrm_gram_skinny: rows must be readable (and zero) up to a multiple of 16 cells: lda >= %lld", (long long)n16);
	NRM_REQUIRE(k_pad >= n && k_pad % SKC == 0 && ldz >= k_pad && ldz % 2 == 0, "nrm_gram_skinny: Z must be padded to a multiple of %d cells", SKC);
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
	NRM_REQUIRE(d_work != nullptr, "nrm_gram_skinny: workspace of nrm_gram_skinny_workspace_bytes() bytes required");
	SkinnySched s;
	s.work = (double*)d_work;
	const int64_t tiles = rows_pad / SKM;
	s.nkt = (int)(k_pad / SKC);
	s.nwg = 2 * g_num_cu_s;
	s.nwg -= s.nwg % 8;
	const int64_t waves = tiles / s.nwg, rem = tiles - waves * s.nwg;
	const int64_t sk = rem;  // < nwg, so a workgroup's unit range spans at most two tiles (two slabs per workgroup)
	s.tiles_sk = (int)sk;
	s.tiles_dp = (int)(tiles - sk);
	s.units_per_wg = (int)((sk * s.nkt + s.nwg - 1) / s.nwg);
	// nz <= 16 used Z rows: one MFMA column tile instead of two (half the matrix-core work: the pass becomes HBM-bound);
	// 17..24: one column tile plus one or two 4-row groups on the 4x4x4 instruction (80 / 96 cycles per step instead of 128)
	const int variant = (nz > 0 && nz <= 16) ? 0 : (nz > 16 && nz <= 20) ? 1 : (nz > 20 && nz <= 24) ? 2 : 3;
#define SK_LAUNCH(TT, NT_, NQ_)                                                                                                        \
	hipLaunchKernelGGL((k_gram_skinny<TT, NT_, NQ_>), dim3((unsigned)s.nwg), dim3(256), 0, st, (const TT*)d_a, rows, n16, lda, d_z, ldz, \
					   d_g, d_ss, s)
	if (a_dtype == NRM_F64) {
		if (variant == 0) SK_LAUNCH(double, 1, 0);
		else if (variant == 1) SK_LAUNCH(double, 1, 1);
		else if (variant == 2) SK_LAUNCH(double, 1, 2);
		else SK_LAUNCH(double, 2, 0);
	} else {
		if (variant == 0) SK_LAUNCH(float, 1, 0);
		else if (variant == 1) SK_LAUNCH(float, 1, 1);
		else if (variant == 2) SK_LAUNCH(float, 1, 2);
		else SK_LAUNCH(float, 2, 0);
	}
#undef SK_LAUNCH
	if (s.tiles_sk > 0) hipLaunchKernelGGL(k_skinny_fixup, dim3((unsigned)s.tiles_sk, (SK_SLAB / 2 + 255) / 256), dim3(256), 0, st, d_g, d_ss, s);
	return nrm_check_launch("k_gram_skinny");
}
