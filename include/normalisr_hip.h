/*
 * normalisr_hip.h -- C ABI of libnormalisr_hip.so: the MI355X (gfx950) implementation of Normalisr's
 * linear-association hot path (covariate residualisation -> Gram contraction -> R^2 -> p-value).
 *
 * The reference (lingfeiwang/normalisr v1.0.0) is pure Python and has no FFI; the seams this ABI
 * replaces are named per function below as file:line under /root/reference/src/normalisr/.
 * Conventions:
 *   - every function returns 0 on success or a negative NRM_E_* code; nrm_last_error() gives the text;
 *   - all matrices are row-major with cells contiguous (association.py:163-170); `ld*` are row pitches
 *     in ELEMENTS; dtype codes: NRM_F32 = 0, NRM_F64 = 1;
 *   - pointers named d_* are DEVICE pointers (HBM) owned by the caller; the library never frees or
 *     retains them past return; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     kernels are enqueued asynchronously on it;
 *   - pointers named h_* are HOST pointers (numpy buffers); those entry points synchronise.
 * No torch / Python types cross this boundary.  Bound from Python with ctypes (normalisr_amd/_lib.py),
 * see INTEGRATION.md for the stub a maintainer of the reference would add.
 */
#ifndef NORMALISR_HIP_H
#define NORMALISR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRM_F32 0
#define NRM_F64 1

#define NRM_OK 0
#define NRM_E_ARG -1      /* bad shape/argument  -> ValueError   (association.py:199-216)  */
#define NRM_E_DEVICE -2   /* HIP runtime failure -> RuntimeError                           */
#define NRM_E_NUMERIC -3  /* non-finite / out-of-range result -> AssertionError (association.py:248-259) */
#define NRM_E_UNSUPPORTED -4  /* the call is outside what a whole-problem host entry covers (nrm_last_error says why): use another path -> NotImplementedError */

/* Tile geometry the padded device buffers must honour (rows multiple of NRM_ROW_TILE, pitch and
 * cell count multiple of NRM_K_TILE). */
#define NRM_ROW_TILE 128
#define NRM_K_TILE 16

/* Number of u-polynomial coefficients of the p-value fast path (see nrm_pvalue_plan). */
#define NRM_PCOEF 20

int nrm_version(void);
const char* nrm_last_error(void);
/* hipGetDeviceCount / hipSetDevice wrappers so that a pure-C host needs no HIP headers. */
int nrm_device_count(int* count);
int nrm_set_device(int device);
/* Result buffers on the host (the caller's numpy arrays, filled where the reference's gather loop fills them,
 * association.py:1005-1034): nrm_host_pin faults the pages of [ptr, ptr+bytes) in with `threads` host threads
 * (0 = choose; contents are preserved) and page-locks the range so device->host copies run at the PCIe rate;
 * nrm_copy_to_host queues such a copy on `stream`; nrm_host_unpin releases the lock (after the stream is synchronised). */
int nrm_host_pin(void* ptr, int64_t bytes, int threads);
int nrm_host_unpin(void* ptr);
int nrm_copy_to_host(void* h_dst, const void* d_src, int64_t bytes, void* stream);
/* Page-locked host memory owned by the library's caller (hipHostMalloc / hipHostFree): the Python host keeps a pool of such
 * blocks behind the numpy result arrays it returns, so that repeated calls neither fault in nor register fresh pages. */
int nrm_host_alloc(void** ptr, int64_t bytes);
int nrm_host_free(void* ptr);
/* a rectangle (rows x row_bytes, pitches in bytes) of a device matrix -> the same rectangle of a host matrix, queued on `stream` */
/* Host side of a symmetric result that arrives by rows: h[0:a, a:b] = h[a:b, 0:a]^T for a row-major matrix of 4- or 8-byte elements
 * (`threads` host threads, 0 = one per 4 hardware threads up to 32).  The numpy-out coex path ships only the rows a..b up to column b over
 * PCIe and mirrors them here while later rows are in flight (the reference mirrors on the host too: association.py:1049-1057). */
int nrm_host_mirror_rows(void* h, int64_t ld_bytes, int elem_bytes, int64_t a, int64_t b, int threads);
int nrm_copy_rect_to_host(void* h_dst, int64_t dst_pitch, const void* d_src, int64_t src_pitch, int64_t row_bytes, int64_t rows, void* stream);
/* Device-side assembly of the padded operand buffers (zero padding, stacking [C; X~] for the streaming path, gathering sums of
 * squares of row chunks): asynchronous on `stream`; pitches and row_bytes in BYTES. */
int nrm_fill_zero(void* d_dst, int64_t bytes, void* stream);
int nrm_fill_i32(void* d_dst, int32_t value, int64_t count, void* stream);  /* count int32 words = value */
int nrm_copy_rows(void* d_dst, int64_t dst_pitch, const void* d_src, int64_t src_pitch, int64_t row_bytes, int64_t rows, void* stream);

/*
 * K1 -- residualise rows against covariates and take their sums of squares.
 * Replaces association.py:224-233 (ccx = dci@(dc@dx.T); dx1 = dx - ccx@dc; mean of squares) and
 * removes the per-tile recomputation (the reference redoes this inside every 500x500 tile).
 *   d_x     (rows, n) input rows, dtype x_dtype, pitch ldx
 *   d_c     (nc, n) fp64 covariates, pitch ldc;  d_dci (nc, nc) fp64 pseudo-inverse of C C^T
 *           (computed on the host in fp64: inv_rank, association.py:4-134); rank = integer rank;
 *           rank == 0 or nc == 0 -> rows are copied unchanged (association.py:224).
 *   d_out   (rows_pad, ldo) fp64 residualised rows, zero-filled for cells >= n and rows >= rows
 *   d_ss    (rows_pad) fp64  sum_k out[i,k]^2   (variance * n; the 0 -> 1 rule is applied later)
 *   d_coef  (rows, nc) fp64 or NULL: the OLS coefficients ccx, needed only for alpha (lowmem=False)
 */
int nrm_residualize(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx,
					const double* d_c, int64_t nc, int64_t ldc, const double* d_dci, int rank,
					double* d_out, int64_t ldo, int64_t rows_pad,
					double* d_ss, double* d_coef, void* stream);

/* K1 with fixed-point output for the integer Gram engine (see nrm_quantize_rows below): the residuals are rounded and cut into
 * digit planes inside K1 (a second sweep over the rows, which are in L2 by then), so the fp64 residuals need not travel through
 * HBM at all: d_out may be NULL.  Needs 16-byte aligned rows and rows_pad % NRM_ROW_TILE == 0; d_q holds
 * nrm_quant_bytes(rows_pad, round_up(n, 16), nslices) bytes, d_exp rows_pad int32.  plane_pitch_bytes != 0: the rows are a
 * block of 32-row groups of a larger quantised matrix (d_q points at the block's first group, the pitch is the larger matrix's:
 * rows that arrive chunk by chunk fill one set of planes).  d_cmax (nc) = max_k |C[c,k]| of every covariate row, or NULL: with it
 * the fixed-point scale of a row comes from the bound max|x| + sum_c |b_c| max|C_c| >= max|residual| and the rows are swept twice
 * instead of three times -- as long as that bound lies within 12x of the residuals' rms (estimated in the first sweep); rows whose
 * mean dwarfs their spread, or whose covariates nearly cancel, are swept for their true maximum so that the overestimate never
 * costs more than a few of the 8 * nslices - 2 bits.
 * d_fix (rows_pad, NRM_FIX_STRIDE) fp64 or NULL: one record per row for K3 (nrm_assoc_sweep* below): the digit sums of the planes
 * whose products the integer engine leaves out (K3 adds the product of the digit means back exactly) and the two numbers of the
 * accuracy guard, c = sqrt(max digit variance / |q|^2) and g = sqrt(n) / (2 |q|); see csrc/nrm_fix.h.  The reference computes this
 * contraction in fp64 (association.py:224-235); the records are what lets the integer engine certify, pair by pair, that its
 * P-values agree with that to the stated tolerance.  Rows of fewer than 2^22 cells. */
#define NRM_FIX_STRIDE 8
int nrm_residualize_q(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx,
					  const double* d_c, int64_t nc, int64_t ldc, const double* d_dci, int rank,
					  double* d_out, int64_t ldo, int64_t rows_pad, double* d_ss, double* d_coef,
					  int nslices, void* d_q, int32_t* d_exp, int64_t plane_pitch_bytes, const double* d_cmax, double* d_fix, void* stream);
/* Profiling aid, not part of the product path: d_stamps = a device buffer of 6 int64 per row of nrm_binnet* (time stamps: start, row loaded,
 * threshold found, mask written; counting passes; spare) that later launches fill; NULL switches it off (tools/time_binnet.py). */
int nrm_binnet_debug_buffer(void* d_stamps);
/* The same with the digit planes cut along the cells into chunks of 32 * chunk_ksteps cells: chunk c is a dense quantised operand
 * of its own (nslices planes of rows_pad / 32 * chunk_ksteps KB) at d_q + c * nrm_quant_bytes(rows_pad, 32 * chunk_ksteps, nslices);
 * all chunks share d_exp; the last chunk is zero padded.  The sharded coex path (normalisr_amd/distributed.py; the N > 1 form of
 * association.py:890-909) sends the chunks to the other GPUs one after another and contracts each as it lands. */
int nrm_residualize_q_chunked(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx,
							  const double* d_c, int64_t nc, int64_t ldc, const double* d_dci, int rank,
							  int64_t rows_pad, double* d_ss, int nslices, void* d_q, int32_t* d_exp,
							  int64_t chunk_ksteps, const double* d_cmax, double* d_fix, void* stream);

/*
 * K2 -- Gram contraction dot[i,j] = sum_k A[i,k] B[j,k] on the fp64 matrix cores
 * (v_mfma_f64_16x16x4_f64, 128x128 workgroup tiles staged through LDS).
 * Replaces np.matmul(dy1, dx1.T) at association.py:234 for ALL tiles of the problem at once.
 *   d_a (m_pad, lda), d_b (n_pad, ldb) fp64, zero padded; m_pad, n_pad multiples of NRM_ROW_TILE;
 *   k_pad multiple of NRM_K_TILE; d_dot (m_pad, ldd) fp64.
 *   symmetric != 0 (coex: d_b == d_a): only tiles on or above the block diagonal are computed and
 *   written (association.py:893-894); the strictly-lower tiles of d_dot are left untouched, and inside
 *   diagonal tiles only 16x16 sub-blocks on or above the diagonal are guaranteed.
 *   m_rows, n_rows: valid (unpadded) row counts (0 = all): 16-row sub-blocks that are pure padding are skipped
 *   and their outputs left unwritten.
 */
int nrm_gram_f64(const double* d_a, const double* d_b, int64_t m_pad, int64_t n_pad, int64_t k_pad,
				 int64_t lda, int64_t ldb, double* d_dot, int64_t ldd, int symmetric,
				 int64_t m_rows, int64_t n_rows, void* d_work, void* stream);
/* The same for the output rows [row0, row1) only (cut at multiples of 8 * NRM_ROW_TILE, or at m_pad): lets the caller
 * pipeline K2 -> K3 -> copy-out band by band, as the reference hands finished tiles to the gather loop
 * (association.py:997-1034).  Consecutive bands on one stream may share d_work. */
int nrm_gram_f64_band(const double* d_a, const double* d_b, int64_t m_pad, int64_t n_pad, int64_t k_pad,
					  int64_t lda, int64_t ldb, double* d_dot, int64_t ldd, int symmetric,
					  int64_t m_rows, int64_t n_rows, int64_t row0, int64_t row1, void* d_work, void* stream);
/* Size of the device scratch nrm_gram_f64 needs in d_work (partial tiles of the stream-K tail; summed in a fixed
 * order, so results are bitwise reproducible).  Independent of the problem size. */
int64_t nrm_gram_workspace_bytes(void);

/*
 * K2, integer engine -- the same contraction computed EXACTLY on the int8 matrix cores from fixed-point operands
 * (csrc/nrm_gram_i8.hip): every row is scaled by a power of two, rounded once to 8 * nslices - 2 bits (nslices = 6: 46 bits,
 * 1.4e-14 of the row's largest entry; 5: 38 bits) and cut into nslices balanced base-256 digits; the digit products are
 * accumulated in int32 without rounding and combined in fp64.  About 2.5x (6 slices) / 3.4x (5) the rate of the fp64 kernel.
 *   nrm_quantize_rows: d_x (rows_pad, ldx) fp64 rows as written by nrm_residualize (rows_pad % 32 == 0 -- % NRM_ROW_TILE for
 *       nrm_gram_i8 operands --, k_pad % 16 == 0, zero padded) -> d_q (nrm_quant_bytes() bytes: digit planes in the kernel's tiled layout) and d_exp (rows_pad) int32 with
 *       x = q * 2^exp.
 *   nrm_gram_i8_band: as nrm_gram_f64_band with quantised operands (d_qb == d_qa, d_eb == d_ea for symmetric problems).
 *       plane_*_bytes: distance between an operand's digit planes, 0 = dense (m_pad / 32 * ceil(k_pad / 32) KB); an operand may be
 *       a block of 32-row groups of a larger quantised matrix: pass the pointer of its first group and the larger matrix's pitch.
 */
int64_t nrm_quant_bytes(int64_t rows_pad, int64_t k_pad, int nslices);
int nrm_quantize_rows(const double* d_x, int64_t rows_pad, int64_t k_pad, int64_t ldx, int nslices, void* d_q, int32_t* d_exp,
					  double* d_fix /* row records as in nrm_residualize_q, or NULL */, int64_t n_cells /* valid cells (<= k_pad), for d_fix */, void* stream);
int nrm_gram_i8_band(const void* d_qa, const int32_t* d_ea, int64_t plane_a_bytes, const void* d_qb, const int32_t* d_eb,
					 int64_t plane_b_bytes, int64_t m_pad, int64_t n_pad, int64_t k_pad, int nslices, double* d_dot, int64_t ldd,
					 int symmetric, int64_t m_rows, int64_t n_rows, int64_t row0, int64_t row1, void* d_work, void* stream);
/* One cell chunk (k_pad = its cells) of a contraction whose operands arrive in chunks: accumulate != 0 adds the chunk's exact
 * partial dot products to d_dot in fp64 instead of overwriting it.  b_block_rows != 0: operand B is n_pad / b_block_rows blocks
 * (cyclically from block b_first of b_count) of a gathered buffer, block b a dense quantised operand of b_block_rows padded rows
 * at d_qb + b * b_block_stride_bytes with its exponents at d_eb + b * b_block_rows: all full partner blocks of a rank in one
 * launch (the N > 1 form of the x0 <= y0 tile loop, association.py:890-909). */
int nrm_gram_i8_chunk(const void* d_qa, const int32_t* d_ea, int64_t plane_a_bytes, const void* d_qb, const int32_t* d_eb,
					  int64_t plane_b_bytes, int64_t m_pad, int64_t n_pad, int64_t k_pad, int nslices, double* d_dot, int64_t ldd,
					  int symmetric, int64_t m_rows, int64_t n_rows, int accumulate, int64_t b_block_rows, int64_t b_block_stride_bytes,
					  int b_first, int b_count, void* d_work, void* stream);

/*
 * P-value plan: host-side constants of p = I_{1-R^2}(dof/2, 1/2) for one dof
 * (scipy.stats.beta.cdf call at association.py:249).  dof is uniform per call for single=0:
 * dof = n_cell - 1 - rank - dimreduce.
 */
typedef struct nrm_pvalue_plan {
	double a;                /* dof / 2                                                   */
	double alpha;            /* a - 1/4                                                   */
	double ln_front;         /* ln( Gamma(a+1/2) / (Gamma(a) sqrt(pi)) )                  */
	double umax;             /* fast path used for -ln(1-R^2) <= umax (0 = never)         */
	double coef[NRM_PCOEF];  /* fast path: p = exp(-alpha u)(erfcx(sqrt(alpha u)) + sqrt(alpha u) sum_j coef[j] u^j) */
} nrm_pvalue_plan;
int nrm_pvalue_plan_init(nrm_pvalue_plan* plan, double dof);
/* The same for `count` dofs (single=1: one dof per grouping, association.py:374): record j = the 24 doubles of the plan
 * for dof[j], written at out + j * pitch (pitch in doubles, >= 24). */
int nrm_pvalue_plan_init_many(const double* dof, int64_t count, double* out, int64_t pitch);
/* The same records from the double-precision form the device builds its plans with (csrc/nrm_pvalue_plan.h, here compiled for the host). */
int nrm_pvalue_plan_fill_many(const double* dof, int64_t count, double* out, int64_t pitch);

/* Elementwise p-values from R^2 (device arrays); kernel-level entry for the table tests. */
int nrm_pvalues_from_r2(const double* d_r2, int64_t count, double dof, double* d_p, void* stream);

/*
 * K3 -- per-pair sweep: R^2 = dot^2/(ssx_i ssy_j) -> p-value, effect size, optional Pearson r and t.
 * Replaces association.py:231-235,248-249 and the gather/symmetrise steps :1037-1057.
 *   d_dot (.., ldd) from nrm_gram_f64;  d_ssx (nx), d_ssy (ny) from nrm_residualize
 *   n_cells, dof as above.  zero sums of squares are replaced by n_cells (variance 0 -> 1, :231-233).
 *   symmetric != 0: coex -- pair (i,j) reads dot[min(i,j), max(i,j)], diagonal outputs are exactly 0
 *                   (association.py:1050-1057); requires nx == ny and d_ssx == d_ssy.
 *   stat_kind: 0 -> covariance x~.y~/n ("dot", association.py:1039,1048); 1 -> gamma = x~.y~/ssx (:234)
 *   outputs (nx, ldo) of dtype out_dtype; d_r / d_t may be NULL (north-star extras: Pearson r, t statistic)
 *   d_flags: int32[4] device counters, incremented for non-finite inputs [0] and R^2 > 1+1e-8 [1]
 *            (the reference's assertions at association.py:248,252); may be NULL when fix_nslices == 0.
 *   fix_nslices != 0 (5 or 6: d_dot came from the integer engine with that many digit planes): d_fixx (nx, NRM_FIX_STRIDE) and
 *            d_fixy (ny, ..) are the row records K1 wrote.  Every dot product first receives the exact correction for the coherent
 *            part of the digit products the engine left out; then, guard_tol > 0, the pair is checked: d_flags[2] counts the pairs
 *            whose P-value the engine's remaining error (a rigorous bound, csrc/nrm_fix.h) could move by more than guard_tol
 *            (relative) -- the host redoes such a call on nrm_gram_f64 -- and d_flags[3] receives the largest error estimate seen
 *            (bits of a float).
 */
int nrm_assoc_sweep(const double* d_dot, int64_t ldd, const double* d_ssx, const double* d_ssy,
					int64_t nx, int64_t ny, int64_t n_cells, double dof, int symmetric, int stat_kind,
					void* d_p, void* d_stat, void* d_r, void* d_t, int out_dtype, int64_t ldo,
					int32_t* d_flags, int fix_nslices, const double* d_fixx, const double* d_fixy, double guard_tol, void* stream);
/* The same for the x rows [row0, row1) (multiples of 64, or nx).  Symmetric problems: the band reads only dot rows
 * < row1 and, once the bands [0, row1) have run in order, output rows [0, row1) are complete (mirrored halves included). */
int nrm_assoc_sweep_band(const double* d_dot, int64_t ldd, const double* d_ssx, const double* d_ssy,
						 int64_t nx, int64_t ny, int64_t n_cells, double dof, int symmetric, int stat_kind,
						 void* d_p, void* d_stat, void* d_r, void* d_t, int out_dtype, int64_t ldo,
						 int32_t* d_flags, int64_t row0, int64_t row1, int fix_nslices, const double* d_fixx, const double* d_fixy,
						 double guard_tol, void* stream);

/* The same for an off-diagonal RECTANGLE of a symmetric (coex) problem: rows [r0, r0 + mx) against columns [c0, c0 + my),
 * c0 + my <= r0; d_dot (mx, ldd) holds that rectangle of the Gram matrix, d_ssx / d_ssy the sums of squares of its rows / columns.
 * Every pair is computed once and written at (r0 + i, c0 + j) AND (c0 + j, r0 + i) of the (ng, ldo) outputs (covariance in d_stat),
 * so a coex whose rows arrive chunk by chunk can finish and ship all pairs of a chunk at once (association.py:1049-1057). */
int nrm_assoc_sweep_mirror(const double* d_dot, int64_t ldd, const double* d_ssx, const double* d_ssy, int64_t mx, int64_t my,
						   int64_t n_cells, double dof, void* d_p, void* d_stat, int out_dtype, int64_t ldo, int64_t r0, int64_t c0,
						   int32_t* d_flags, int fix_nslices, const double* d_fixx, const double* d_fixy, double guard_tol, void* stream);

/*
 * alpha[i,j,c] = by[j,c] - gamma[i,j] * bx[i,c]  (association.py:238-243), fp64 coefficients in, out_dtype out.
 * The reference computes alpha from gamma whatever return_dot says (return_dot only rescales the returned
 * statistic afterwards, association.py:1044-1048), so d_stat may hold either form of K3's output:
 *   stat_kind 1: d_stat = gamma;  stat_kind 0: d_stat = covariance x~.y~/n, turned back into gamma with
 *   d_ssx (nx, sums of squares from nrm_residualize; 0 -> n_cells as in K3) and n_cells.
 */
int nrm_alpha(const void* d_stat, int stat_dtype, int64_t ldg, int stat_kind, const double* d_ssx, int64_t n_cells,
			  const double* d_bx, const double* d_by, int64_t nx, int64_t ny, int64_t nc, void* d_alpha, int out_dtype,
			  void* stream);

/*
 * K2s + sweep -- streaming path for de with few design rows (nx + nc <= 32): HBM-bound, every expression
 * value is read once, raw (fp32/fp64), never materialised as an fp64 residual (see csrc/nrm_gram_skinny.hip).
 *   nrm_gram_skinny:  G[y,:] = sum_k Y[y,k] Z[:,k] (rows_pad, 32) and ss[y] = sum_k Y[y,k]^2, with
 *       d_z (32, ldz) fp64 = [C (nc rows); X~ (nx rows, residualised by nrm_residualize); zero rows],
 *       zero padded to k_pad cells (multiple of 128); rows_pad multiple of 256.  d_a rows must be 16-byte aligned (lda % (16/itemsize) == 0).
 *   nrm_de_small_sweep: |y~|^2 = ss - (y C^T) dci (C y^T), y~.x~ = y.x~  ->  R^2, p, gamma|cov (association.py:226-235,249);
 *       d_ssy (ny) receives |y~|^2; d_by (ny, nc) or NULL receives the OLS coefficients ccy (for alpha).
 */
/*
 * The OLS products a = x C^T of the design rows of the streaming de path (first half of association.py:224-227 for dx): rows <= 32
 * design rows against nc <= 32 covariate rows, spread along the cells, partial sums added in a fixed order.
 *   d_ga (rows, 32): a[r][c] at d_ga[r * 32 + c]; the last covariate in column 31 when const_last != 0 (the layout
 *   nrm_residualize_wide reads).  d_work: nrm_design_products_workspace_doubles(rows, n_cells) doubles.
 */
int64_t nrm_design_products_workspace_doubles(int64_t rows, int64_t n_cells);
int nrm_design_products(const void* d_x, int x_dtype, int64_t rows, int64_t n_cells, int64_t ldx, const double* d_c, int64_t nc, int64_t ldc,
						double* d_ga, double* d_work, int const_last, void* stream);

/* Residualise <= 32 rows with the work spread along the cells (used for the design rows of the streaming path):
 * d_ga (rows, 32) holds x C^T in its first nc columns (from nrm_gram_skinny against Z = [C; 0]); out (rows, ldo) fp64,
 * zero padded up to ldo; d_ss (rows) sums of squares; d_coef (rows, nc) or NULL the OLS coefficients.  A row the covariates explain to twenty digits
 * (|x~|^2 < 1e-22 |x|^2) is explained exactly: its row of d_out is cleared, its sum of squares 0 (variance 0 -> 1, P = 1). */
int nrm_residualize_wide(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx, const double* d_c, int64_t nc,
						 int64_t ldc, const double* d_ga, const double* d_dci, int rank, double* d_out, int64_t ldo,
						 double* d_ss, double* d_coef, double* d_work /* 64 * ceil(ldo/1024) doubles */,
						 int const_last /* != 0: the LAST covariate is the constant row: its product sits in column 31 of d_ga */, void* stream);
int nrm_gram_skinny(const void* d_a, int a_dtype, int64_t rows, int64_t n, int64_t lda, const double* d_z, int64_t ldz,
					int64_t k_pad, double* d_g, double* d_ss, int64_t rows_pad, int64_t nz /* used rows of Z (<= 32); <= 16 selects the half-width variant */,
					double const_row_value /* != 0: G[:, 31] = value * sum_k Y[y,k], the product with a constant covariate row (the
					intercept) that the caller left out of Z: a vector-ALU sum instead of a 4-row matrix-core group; 0: none */,
					void* d_work, void* stream);
int64_t nrm_gram_skinny_workspace_bytes(void);  /* scratch for d_work (deterministic combination of partial pieces) */
int nrm_de_small_sweep(const double* d_g, const double* d_ssraw, const double* d_dci, int64_t nc, int rank, const double* d_ssx,
					   int64_t nx, int64_t ny, int64_t n_cells, double dof, int stat_kind, void* d_p, void* d_stat, void* d_r,
					   void* d_t, int out_dtype, int64_t ldo, double* d_ssy, double* d_by, int32_t* d_flags,
					   int const_last /* as above: covariate nc-1 in column 31, the design rows from column nc-1 on */, void* stream);

/*
 * single=4 sweep (competition-aware DE, association.py:421-576 in closed form).
 * Inputs from one multiple regression of every gene on A = [dx; dc] (m = nx + nc rows):
 *   d_bt  (ny, ldb) fp64: Bt[y,k] = coefficient of row k of A for gene y      (B = (A A^T)^-1 A Y^T)
 *   d_pt  (ny, ldb) fp64: Pt[y,k] = A_k . y                                   (prody^T, association.py:952-967)
 *   d_yy  (ny) fp64: sum_k y^2 (association.py:968);  d_dxx (nx) fp64: 1/(n (AA^T)^-1_ii) = variance of x_i
 *   unexplained by all other rows (association.py:539-540).
 * Outputs (nx, ldo) of out_dtype: p, gamma (association.py:550) or gamma*varx when return_dot, vary (:548).
 * dof = n - 1 - (m - 1) - dimreduce is uniform for full-rank A A^T.  d_work: ny doubles of scratch.
 */
int nrm_single4_sweep(const double* d_bt, const double* d_pt, int64_t ldb, const double* d_yy, const double* d_dxx,
					  int64_t nx, int64_t ny, int64_t m, int64_t n_cells, double dof, int return_dot,
					  void* d_p, void* d_stat, void* d_vary, int out_dtype, int64_t ldo, double* d_work, int32_t* d_flags,
					  void* stream);
/* single=4 with the products Y~ X~^T from the integer Gram engine (normalisr_amd/single4.py: the regression is taken on rows
 * residualised against the covariates -- Frisch-Waugh -- so that K1's digit planes, K2's exact contraction and the row records of
 * nrm_residualize_q serve it as they serve single=0; replaces association.py:926-980,421-576 for full-rank designs).
 *   nrm_gram_i8_fix_dot: adds the engine's exact mean-product correction (csrc/nrm_fix.h) to a raw dot matrix in place, from the row
 *       records of its rows and of its columns -- what K3 does inside its sweeps, for callers that need the products themselves.
 *   nrm_single4_sweep_guarded: nrm_single4_sweep plus the accuracy guard: d_fix_y (ny, NRM_FIX_STRIDE) the genes' records, d_kappa (nx)
 *       = sum_j |x~_j| |N_ji| / sqrt(N_ii), cstar / gstar the largest c and g among the design rows' records; a pair whose P-value the
 *       products' error bound could move by more than `budget` (relative) is counted in d_flags[2], the largest estimate kept in
 *       d_flags[3] (float bits); d_flags has 4 entries.  The caller redoes the genes with hits (or the whole call) on the fp64 Gram kernel. */
int nrm_gram_i8_fix_dot(double* d_dot, int64_t ldd, const double* d_fix_rows, const double* d_fix_cols, int64_t rows, int64_t cols,
						int nslices, int64_t n_cells, void* stream);
int nrm_single4_sweep_guarded(const double* d_bt, const double* d_pt, int64_t ldb, const double* d_yy, const double* d_dxx,
							  int64_t nx, int64_t ny, int64_t m, int64_t n_cells, double dof, int return_dot, void* d_p,
							  void* d_stat, void* d_vary, int out_dtype, int64_t ldo, double* d_work, int32_t* d_flags,
							  const double* d_fix_y, const double* d_kappa, double cstar, double gstar, int nslices, double budget,
							  int32_t* d_gene_hits /* (ny) zeroed by the caller, or NULL: 1 for every gene with a counted pair */, void* stream);

/*
 * The small steps around single=4's on-device inverse of M~ = X~ X~^T (normalisr_amd/single4.py: Newton-Schulz iteration X <- X (2 I - M X), its
 * two products per step on nrm_gram_f64; replaces the per-grouping SVDs of association.py:527-528 for full-rank designs).  All matrices
 * (nxp, nxp) fp64 row-major, nxp a multiple of 128 >= nx; every reduction in a fixed order.
 *   nrm_spd_prepare: d_m (.., ldm) as a symmetric nrm_gram_f64 launch leaves it (entries i <= j valid) -> d_mp, the full symmetric matrix,
 *       its padding rows carrying mean(diag) on the diagonal; d_scal[0] = ||M||_1, d_scal[1] = mean(diag); d_work: 2 nxp doubles.
 *   nrm_spd_start: d_x = diag(1 / M_ii) (diagonal != 0) or I / ||M||_1.
 *   nrm_spd_transpose_residual: d_tt = d_t^T, d_res[0] = ||I - d_t||_F; d_work: (nxp / 32)^2 doubles.
 *   nrm_spd_update: d_x = 2 d_x - d_xt.
 *   nrm_spd_finish: d_n = (d_x + d_x^T) / 2 inside nx x nx, 0 in the padding; d_small (3, nx) = diag(N), sum_j |N_ij| sqrt(d_ss[j]), sum_j |N_ij|.
 */
int nrm_spd_prepare(const double* d_m, int64_t ldm, int64_t nx, int64_t nxp, double* d_mp, double* d_scal, double* d_work, void* stream);
int nrm_spd_start(const double* d_mp, int64_t nxp, int diagonal, const double* d_scal, double* d_x, void* stream);
int nrm_spd_transpose_residual(const double* d_t, int64_t nxp, double* d_tt, double* d_res, double* d_work, void* stream);
int nrm_spd_update(double* d_x, const double* d_xt, int64_t count, void* stream);
int nrm_spd_finish(const double* d_x, int64_t nx, int64_t nxp, const double* d_ss, double* d_n, double* d_small, void* stream);
/* What the single=4 sweep needs of N~ (d_small (3, nx) of nrm_spd_finish) without the host: d_dxx[i] = 1 / (n_cells N~_ii) (association.py:539-540 in closed form),
 * d_varx the same with the reference's 0 -> 1 (:546-547); d_flags[5] (int32[8]) += design rows whose N~_ii is not a positive finite number. */
int nrm_single4_design_scalars(const double* d_small, int64_t nx, int64_t n_cells, double* d_dxx, double* d_varx, int32_t* d_flags, void* stream);

/*
 * single=1 sweep (every grouping tested on its own subset of cells, association.py:263-390).
 *   d_g  (ny, ldg): Gram of the expression rows with the masked rows W_i = [1_Si C (nc rows); 1_Si x_i], i = 0..nx-1,
 *        column i*(nc+1)+c;  d_g2 (ny, ldg2): Gram of the squared expression rows with the masks 1_Si.
 *   d_info (nx, info_pitch) fp64 per grouping: [ns_i, vx_i, the 24 doubles of struct nrm_pvalue_plan, ccx_i (nc), M_i^+ (nc*nc)]
 *        prepared on the host (inv_rank of C_S C_S^T stays on the host: integer rank).
 * Outputs (nx, ldo): p, gamma (or gamma*vx when return_dot), vary; d_alpha (nx, ny, nc) or NULL.
 */
int nrm_single1_sweep(const double* d_g, int64_t ldg, const double* d_g2, int64_t ldg2, const double* d_info, int64_t info_pitch,
					  int64_t nc, int64_t nx, int64_t ny, int return_dot, void* d_p, void* d_stat, void* d_vary, void* d_alpha,
					  int out_dtype, int64_t ldo, int32_t* d_flags, void* stream);

/*
 * The same single=1 statistics for designs with entries >= 0 (gRNA incidence), without the masked Gram contractions and without a
 * transposed copy of the expression matrix (association.py:263-390,911-925).  Every cell has a code (int32): a cell where every grouping
 * is 0 is NRM_S1_COMMON (-2); a cell where exactly one grouping is not 0 carries its position (>= 0) in the list of such cells ordered
 * by grouping (d_seg[i] .. d_seg[i+1] are the positions of grouping i); any other cell is NRM_S1_SKIP (-1).
 *   nrm_single1_stream reads d_y (ny, ldy) once, where it lies (y_dtype), with d_c (nc, ldc) fp64 covariates, and leaves
 *     d_common (nc + 1, ny) fp64: rows c < nc = the sums of y C_c over the common cells, row nc = the sum of y^2 over them;
 *     d_ye (cells with a position, ldye) in y_dtype: the expression values at those cells, transposed (ldye: a multiple of 8, >= ny;
 *     64-byte aligned; columns ny .. ldye are scratch).  nc <= 32 (more than 8 covariates: further passes over d_y).
 *   nrm_single1_cells finishes every (grouping, gene) pair from d_common, d_ye, d_ce (cells with a position, nc) fp64 covariates and
 *     d_xe (the same cells) fp64 own value of the grouping; d_info, outputs and flags as nrm_single1_sweep.
 */
#define NRM_S1_COMMON (-2)
#define NRM_S1_SKIP (-1)
int nrm_single1_stream(const void* d_y, int y_dtype, int64_t ldy, const double* d_c, int64_t ldc, int64_t nc, const int32_t* d_code,
					   int64_t n, int64_t ny, double* d_common, void* d_ye, int64_t ldye, void* stream);
int nrm_single1_cells(const void* d_ye, int y_dtype, int64_t ldye, const double* d_ce, const double* d_xe, const int64_t* d_seg,
					  const double* d_common, const double* d_info, int64_t info_pitch, int64_t nc, int64_t nx, int64_t ny, int return_dot,
					  void* d_p, void* d_stat, void* d_vary, void* d_alpha, int out_dtype, int64_t ldo, int32_t* d_flags, void* stream);
/* The groupings' own statistics over their own cells (association.py:350-364): d_seg (nx + 1) delimits grouping i's entries of d_cells (cell
 * indices, int64) / d_xe (its value there); d_out (nx, nc (nc + 1) / 2 + nc + 1): the sums of C_c C_d (c <= d, row by row), of C_c x, of x x.  nc <= 8. */
int nrm_single1_group_stats(const int64_t* d_seg, const int64_t* d_cells, const double* d_xe, const double* d_c, int64_t ldc, int64_t nc, int64_t nx,
							double* d_out, void* stream);
/* The grouping side of association_test_2's loop body (association.py:350-374) on the device, a lane per grouping: M_i = (the shared cells' covariate
 * Gram matrix, d_gram_part of nrm_single1_select added up in block order) + (the grouping's own, d_gs of nrm_single1_group_stats); its pseudo-inverse and
 * INTEGER rank by the rule of inv_rank (association.py:77-80; the Jacobi code the host's nrm_small_pinv runs); ccx_i, vx_i (0 -> 1), dof_i = ns_i - 1 -
 * rank_i - dimreduce and the P-value plan for it -- written as the records nrm_single1_cells reads (d_info (nx, info_pitch >= 26 + nc + nc nc)) -- and
 * d_varx (nx) fp64.  What the reference asserts or raises there is counted into d_flags (int32[8], zeroed by the caller; [0], [1] are the sweep's):
 * [2] groupings with a single value on their selected cells (:917-918), [3] groupings with dof <= 0, [4] groupings whose M_i is not finite.  nc <= 8. */
int nrm_single1_group_info(const double* d_gs, const double* d_gram_part, const double* d_rowinfo, const int64_t* d_sel_info, int64_t nc, int64_t nx,
						   int dimreduce, double* d_info, int64_t info_pitch, double* d_varx, int32_t* d_flags, void* stream);

/*
 * binnet -- binarise a (ng, ng) co-expression P-value matrix at a per-row Benjamini-Hochberg q-value cutoff
 * (reference binnet.py:134-173 with bh :77-131; the consumer of coex's p-matrix).  d_out (ng, ldo) bytes 0/1, diagonal 0;
 * *d_total receives the number of selected entries (0 -> the reference raises "Empty binary network");
 * d_flags[0] counts rows with entries outside [0,1] / non-finite (reference assertions :151-152).
 */
int nrm_binnet(const void* d_p, int p_dtype, int64_t ng, int64_t ldp, double qcut, unsigned char* d_out, int64_t ldo,
			   unsigned long long* d_total, int32_t* d_flags, void* stream);
/* The same for a block of gene rows [row0, row0 + rows) of the (ng, ng) matrix: d_p (rows, ldp) and d_out (rows, ldo) hold only
 * that block (row i of the block is gene row0 + i; its diagonal entry is column row0 + i).  Rows are independent in the reference
 * (binnet.py:159 maps bh over the rows), so a GPU that owns a row block of a sharded coex binarises it without the other blocks;
 * *d_total counts this block's selected entries (the caller sums the blocks before the "Empty binary network" test). */
int nrm_binnet_rows(const void* d_p, int p_dtype, int64_t rows, int64_t ng, int64_t ldp, int64_t row0, double qcut,
					unsigned char* d_out, int64_t ldo, unsigned long long* d_total, int32_t* d_flags, void* stream);

/*
 * normvar (reference norm.py:131-289): per-gene weighted covariate removal, e_gk = w_k^wt_g.
 *   nrm_normvar_weights: U = e^2, V = e^2 * y  (fp64, (rows_pad, ldo), zero padded) for the two Gram contractions
 *       M_g = U P^T (P = products C_c*C_c') and a_g = V C^T on nrm_gram_f64;  s1 = sum_k y e, s2 = sum_k (y e)^2.
 *       d_lnw (n) = ln w, d_wt (rows) = wt.
 *   nrm_normvar_apply:   out_gk = scale_g * e_gk * (y_gk - sum_c b_gc C_ck),  b_g = M_g^+ a_g from the host (integer rank).
 */
int nrm_normvar_weights(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, const double* d_lnw, const double* d_wt,
						double* d_u, double* d_v, int64_t ldo, int64_t rows_pad, double* d_s1, double* d_s2, void* stream);
int nrm_normvar_apply(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, const double* d_lnw, const double* d_wt,
					  const double* d_c, int64_t nc, int64_t ldc, const double* d_b, const double* d_scale, void* d_out, int out_dtype,
					  int64_t ldo, int32_t* d_flags /* int32[4] or NULL: [1] += waves that wrote a non-finite value (norm.py:286) */, void* stream);
/* d_out[i] = exp(d_x[i]) as the normvar kernels compute their per-gene cell weights w_k^wt_g = exp(wt_g ln w_k) (norm.py:245): a table of 2^(j/64) and a
 * polynomial of degree 5 instead of the library's exp (csrc/nrm_normvar.hip: nv_exp).  A probe for the tests. */
int nrm_normvar_exp_probe(const double* d_x, int64_t count, double* d_out, void* stream);
/* Round 5: b_g, scale_g and the integer rank of every gene WITHOUT leaving the device, for 1 .. nrm_normvar_device_covariates() covariates: one pass
 * over d_y sums the per-gene moments (M_g = sum_k e_gk^2 C_k C_k^T, a_g = sum_k e_gk^2 y_gk C_k, sum y e, sum (y e)^2) in registers -- no U, V, no Gram
 * launches --, a thread per gene takes M_g^+ by the rank rule of inv_rank (association.py:77-80; the Jacobi iteration of nrm_small_pinv: same integer
 * ranks) and the variance-keeping scale (norm.py:248-259; keepvar = 0: 1).  d_mom: rows (nc (nc + 1) / 2 + nc + 2) doubles of scratch; d_flags[0] +=
 * genes of rank 0 (norm.py:158-159 raises for them).  Then nrm_normvar_apply. */
int64_t nrm_normvar_device_covariates(void);
int nrm_normvar_solve(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, const double* d_lnw, const double* d_wt, const double* d_c, int64_t nc,
					  int64_t ldc, double tol, int keepvar, double* d_mom, double* d_b, double* d_scale, int64_t* d_rank, int32_t* d_flags, void* stream);
/* normvar1 with explicit per-gene cell weights w2 (rows, ldw) (norm.py:150-153: row g is residualised against dc * w2[g]):
 * out_gk = y_gk - w2_gk * sum_c b_gc C_ck, with b_g = (sum_k w2_gk^2 C_k C_k^T)^+ (sum_k w2_gk y_gk C_k) from the host. */
int nrm_normvar_apply_w2(const void* d_y, int y_dtype, int64_t rows, int64_t n, int64_t ldy, const double* d_w2, int64_t ldw,
						 const double* d_c, int64_t nc, int64_t ldc, const double* d_b, void* d_out, int out_dtype, int64_t ldo, void* stream);

/*
 * Whole-problem host entry (numpy in / numpy out): the seam association_tests(dx, dy, dc, ...)
 * -> (p, dot|gamma, alpha|None, varx|None, vary) at association.py:761-771,1093 for single=0.
 * All pointers are HOST buffers owned by the caller.  h_dy == NULL means dy = dx (coex).
 *   h_dci (nc,nc) fp64 and rank from the host inv_rank;  dimreduce int (association.py:213).
 *   h_p, h_stat (nx,ny) of dtype out_dtype; h_varx (nx) / h_vary (ny) of out_dtype (h_varx may be
 *   NULL); h_alpha (nx,ny,nc) or NULL (lowmem); h_r / h_t optional extras or NULL.
 *   return_dot: 1 -> covariance, 0 -> gamma (association.py:769).
 * Runs on the current device (nrm_set_device) and synchronises before returning.
 */
/* Frees the device scratch nrm_association_tests_host keeps between calls (it is reused best-fit; calls are
 * serialised per process). */
int nrm_release_cache(void);
/* Verdict of the integer engine's accuracy guard for the last nrm_association_tests_host call of this thread: *hits = pairs it could
 * not certify on the integer pass (> 0: the call was redone on the fp64 Gram kernel before returning), *worst = largest relative
 * error estimate of a P-value among the pairs it looked at.  NRM_I8_GUARD_TOL sets the tolerance (default 2.5e-7; 0 = no guard). */
int nrm_last_guard(int64_t* hits, double* worst);
int nrm_association_tests_host(const void* h_dx, int x_dtype, int64_t nx,
							   const void* h_dy, int y_dtype, int64_t ny,
							   const void* h_dc, int c_dtype, int64_t nc, int64_t n_cells,
							   const double* h_dci, int rank, int dimreduce, int return_dot,
							   void* h_p, void* h_stat, void* h_alpha, void* h_varx, void* h_vary,
							   void* h_r, void* h_t, int out_dtype);

/* (Round 5) nrm_association_tests_host streams the raw expression rows (nrm_gram_skinny, as the Python engine does) when dy is given and nx + nc <= 32 --
 * case-control DE, BASELINE configs[2]; NRM_DEBUG="de_path=general" keeps K1 + K2 -- and takes the sparse-design path below by itself when dy is given and the design matrix qualifies -- at most
 * 1/16 of its entries set, >= 32 design rows, >= 64 expression rows, >= 2048 cells, nx n >= 2^22, <= 32 covariates; NRM_DE_SPARSE=0 switches that
 * off, =force takes it whatever the size -- and hands calls whose rows are too close to the span of the covariates back to its dense fp64 path.
 *
 * The other two CLI methods of `normalisr de` at the same seam (association_tests(..., single=1 | 4), association.py:911-980), numpy buffers in
 * and out, no torch.  Outputs as the reference returns them: h_p, h_stat (gamma, or gamma * varx when return_dot), h_vary (nx, ny), h_varx (nx),
 * h_alpha (nx, ny, nc) or NULL -- all of out_dtype.  NRM_E_UNSUPPORTED (nrm_last_error says why) for calls outside what the entry covers:
 *   nrm_association_tests_single1_host (`-m single`, association.py:263-390,911-925): design entries >= 0 of which at most a quarter are set,
 *     at most 32 covariates (the gRNA incidence of a screen; the package's masked-Gram path takes the rest);
 *   nrm_association_tests_single4_host (`-m covariate`, association.py:421-576,926-980): the closed form for full-rank designs -- rank == nc
 *     (h_dci, rank from the host's inv_rank of C C^T) and A A^T certified full rank at `tol` (association.py:77) from norms at hand; a sparse design
 *     goes through the sparse-design kernels, any other through nrm_residualize + nrm_gram_f64.
 * nrm_binnet_host (binnet.py:134-173): h_p (ng, ng) -> h_net (ng, ng) bytes 0 / 1, *total = selected entries (0: the reference raises).
 */
int nrm_association_tests_single1_host(const void* h_dx, int x_dtype, int64_t nx, const void* h_dy, int y_dtype, int64_t ny, const void* h_dc, int c_dtype,
									   int64_t nc, int64_t n_cells, int dimreduce, int return_dot, void* h_p, void* h_stat, void* h_alpha, void* h_varx,
									   void* h_vary, int out_dtype);
int nrm_association_tests_single4_host(const void* h_dx, int x_dtype, int64_t nx, const void* h_dy, int y_dtype, int64_t ny, const void* h_dc, int c_dtype,
									   int64_t nc, int64_t n_cells, const double* h_dci, int rank, int dimreduce, int return_dot, double tol, void* h_p,
									   void* h_stat, void* h_alpha, void* h_varx, void* h_vary, int out_dtype);
int nrm_binnet_host(const void* h_p, int p_dtype, int64_t ng, double qcut, unsigned char* h_net, int64_t* total);
/* (Round 6) What the package needs to follow the reference's per-grouping algorithm of single=4 without torch where the closed form of
 * nrm_association_tests_single4_host does not apply (rank-deficient A A^T, mpc / method / qr, dy=None): the Gram matrices association.py:926-968 forms with
 * numpy.matmul, on the fp64 Gram kernel -- h_out (ra, rb) fp64 = A B^T over n cells (h_b == NULL: B = A), optionally the rows' sums of squares -- and the
 * P-value function of association.py:563 for host arrays.  The small pseudo-inverses of the algorithm stay in numpy (normalisr_amd/single4.py). */
int nrm_gram_host(const void* h_a, int a_dtype, int64_t ra, const void* h_b, int b_dtype, int64_t rb, int64_t n, double* h_out, double* h_ssa, double* h_ssb);
int nrm_pvalues_host(const double* h_r2, int64_t count, double dof, double* h_p);
/* (Round 5) normvar's expression side at the same kind of seam (norm.py:166-289; `normalisr normvar`): h_y (rows, n) fp32 / fp64, h_lnw (n) = ln w, h_wt (rows), h_c (nc, n)
 * fp64 -> h_out (rows, n) of out_dtype = gene g times w^wt_g, the covariates C w^wt_g removed, the variance kept (keepvar != 0, norm.py:248-259); tol: the rank rule of
 * inv_rank (association.py:77).  1 .. nrm_normvar_device_covariates() covariates entirely on the device; up to 32 through two launches of the fp64 Gram kernel and the host's
 * threaded Jacobi stack (round 6; NRM_E_UNSUPPORTED beyond 32).  *zero_rank = genes whose covariates have rank 0 (the
 * reference raises RuntimeError, norm.py:158-159; h_out is not written then); NRM_E_NUMERIC for non-finite results (norm.py:286).  The covariates' own scaling
 * (norm.py:261-273) is element-wise on (nc, n) and stays with the caller. */
int nrm_normvar_host(const void* h_y, int y_dtype, int64_t rows, int64_t n, const double* h_lnw, const double* h_wt, const double* h_c, int64_t nc, double tol,
					 int keepvar, void* h_out, int out_dtype, int64_t* zero_rank);
/* eigenvalues (ascending) of a small symmetric matrix, n <= 32 (host only) */
int nrm_small_eigvals(const double* m, int64_t n, double* w);

/*
 * de with a SPARSE design matrix (a CRISPR screen's gRNA incidence), association.py:224-235 without the dense contraction: a residual is
 * orthogonal to the covariates, so x~_i . y~ = x_i . y - (y C^T) . b_i and |y~|^2 = |y|^2 - (y C^T) . b_y -- the raw expression rows are
 * read ONCE, and of each only the values at the cells where a design row is not zero are added up.
 *   d_y (ny, ldy) raw expression rows (y_dtype); d_common (nc + 1, ny) fp64 their products with the covariates and (row nc) sums of
 *   squares, as nrm_single1_stream leaves them when every cell has the code NRM_S1_COMMON; d_dci (nc, nc) the pseudo-inverse of C C^T,
 *   nc <= nrm_de_sparse_max_covariates() (nc = 0: no covariates, or covariates of rank 0).
 *   The design matrix as lists: its rows are slots 0 .. 64 * ngroups - 1 (d_slot2x: slot -> design row, -1 for an empty slot); the cells
 *   are cut into chunks of nrm_de_sparse_chunk().  In every chunk the slots are dealt anew to the 64 * ngroups POSITIONS of the workgroup's
 *   lanes, d_sig[c * 64 * ngroups + p] = the slot position p gathers for in chunk c (a permutation inside every block of 1024 positions:
 *   one pass of the kernel; sorted by the slots' number of entries in the chunk, so that the 64 lists a wave walks in step are equally
 *   long).  For chunk c and the 64 positions of group g, d_w[c * ngroups + g] is the length of the longest list among them rounded up to a
 *   multiple of 8, and d_base[c * ngroups + g] the start of their block in d_ell: entry j of position 64 g + l at
 *   d_base[..] + (64 (j / 8) + l) 8 + j % 8 (8 consecutive entries of a position side by side: one 16-byte load) = the offset of the cell
 *   inside its chunk, or nrm_de_sparse_chunk() for padding; d_ellv: the entries' values likewise (fp64), or NULL
 *   when every entry is 1.  d_bx (design rows, ldb) the design rows' coefficients b_i from K1.
 * Out: d_dot[i * ldd + y] = x~_i . y~_y (by_gene != 0: d_dot[y * ldd + i], the layout single=4 reads); d_ssy (ny) = |y~|^2; d_coefy (ny, nc)
 *   = b_y or NULL; d_flags (int32[4], as nrm_assoc_sweep's) or NULL: [2] counts the expression rows whose residual is so small a part
 *   of the row (|y~|^2 < 1e-4 |y|^2) that these differences have lost four digits -- redo such a call on nrm_residualize + nrm_gram_f64.
 *   Then nrm_assoc_sweep as for K2's output.
 */
int64_t nrm_de_sparse_chunk(void);
int64_t nrm_de_sparse_max_covariates(void);
/*   d_ct != NULL (round 5): the rows' products with the covariates and their sums of squares are taken INSIDE the kernel, on the fp64 matrix
 *   cores, from the chunk of rows it holds in LDS anyway -- ONE pass over the expression matrix -- and left in d_common ((nc + 1, ny), then
 *   an output; nrm_single1_stream need not run).  d_c (nc, ldc) the covariates; const_idx: a covariate that is constant (the intercept, value
 *   const_val) or -1 -- it needs no operand; at most nrm_de_sparse_fused_covariates() others; d_ct: nrm_de_sparse_ct_doubles() doubles of scratch. */
int64_t nrm_de_sparse_fused_covariates(void);
int64_t nrm_de_sparse_ct_doubles(int64_t n, int64_t nc, int64_t const_idx);
int nrm_de_sparse(const void* d_y, int y_dtype, int64_t ny, int64_t n, int64_t ldy, double* d_common, int64_t nc, const double* d_dci,
				  const int16_t* d_ell, const double* d_ellv, const int64_t* d_base, const int32_t* d_w, const int32_t* d_sig, int64_t ngroups,
				  const int32_t* d_slot2x, const double* d_bx, int64_t ldb, double* d_dot, int64_t ldd, int by_gene, double* d_ssy, double* d_coefy,
				  int32_t* d_flags, const double* d_c, int64_t ldc, int64_t const_idx, double const_val, double* d_ct, void* stream);

/*
 * The design rows' own statistics from their entries (association.py:224-230 for a sparse design row): d_row_ptr (nx + 1), d_cells (int32),
 * d_vals (fp64, or NULL: every entry 1) list the entries of design row i at [d_row_ptr[i], d_row_ptr[i + 1]); d_c (nc, ldc), d_dci as above.
 * d_ss (nx) = |x~_i|^2 = |x_i|^2 - (x_i C^T) . b_i, d_coef (nx, nc) = b_i.  d_flags (int32[4]) or NULL: [2] counts the design rows whose
 * residual is so small a part of the row (|x~|^2 < 1e-4 |x|^2: a gRNA that all but coincides with a covariate) that this difference, and the
 * products nrm_de_sparse forms with the row, have lost four digits -- the same hand-back to nrm_residualize + nrm_gram_f64 as on the
 * expression side.
 */
int nrm_design_stats(const int64_t* d_row_ptr, const int32_t* d_cells, const double* d_vals, const double* d_c, int64_t ldc, int64_t nc,
					 const double* d_dci, int64_t nx, double* d_ss, double* d_coef, int32_t* d_flags, void* stream);

/*
 * The lists above built from the design matrix itself, by kernels of the library (csrc/nrm_design_lists.hip) -- the design side of
 * association.py:224-235 for a sparse design, and what association.py:914-918 selects cells from.  d_x (nx, ldx) the design matrix in HBM
 * (NRM_F32 / NRM_F64); nslots = nx rounded up to a multiple of 64; nch = ceil(n / nrm_de_sparse_chunk()) chunks of cells.
 *   nrm_design_count: ONE pass over d_x.  d_cnt (nch, nslots) int32: low 16 bits = entries (values != 0, NaN included) of design row `slot` in
 *     chunk c, above them bits NRM_DESIGN_* describing those entries; d_info (int64[8]) is zeroed by the call.
 *   nrm_design_plan: from the counts -- d_row_ptr (nx + 1) the CSR offsets, d_coff (nch, nslots) int32 the entries of a row before each chunk,
 *     d_slot2x (nslots) or NULL; with d_sig != NULL also the dealing d_sig / d_pos (nch, nslots: position -> slot and slot -> position, sorted
 *     by the entries of the chunk inside every block of 1024 slots, most first, ties in row order), the widths d_w (nch, nslots / 64) and
 *     offsets d_base of the ELL blocks as nrm_de_sparse reads them.  d_info[0] = entries in all, d_info[1] = entries of the ELL form, padding included,
 *     d_info[2] = the NRM_DESIGN_* bits of all entries.
 *   nrm_design_fill: the second pass over d_x writes the entries (cells ascending inside a row; no atomics: the same lists run to run) into
 *     d_cells / d_row_vals (CSR; d_cells may be NULL) and d_ell / d_ellv (ELL, padding included; d_ell may be NULL); binary != 0: every entry
 *     is 1 (d_info[2] has no NRM_DESIGN_NOTONE) and the value arrays are not written.
 * The caller reads d_info (one small copy) between plan and fill to size d_cells (d_info[0]) and d_ell (d_info[1]).
 */
#define NRM_DESIGN_NOTONE 1     /* an entry that is neither 0 nor 1 */
#define NRM_DESIGN_NEG 2        /* an entry < 0 */
#define NRM_DESIGN_GT1 4        /* an entry > 1 */
#define NRM_DESIGN_HAS1 8       /* an entry == 1 */
#define NRM_DESIGN_NAN 16       /* a NaN (an infinite entry sets NRM_DESIGN_GT1 or NRM_DESIGN_NEG) */
int nrm_design_count(const void* d_x, int x_dtype, int64_t nx, int64_t n, int64_t ldx, int32_t* d_cnt, int64_t nslots, int64_t* d_info, void* stream);
int nrm_design_plan(const int32_t* d_cnt, int64_t nx, int64_t n, int64_t nslots, int32_t* d_sig, int32_t* d_pos, int32_t* d_w, int64_t* d_base,
					int64_t* d_row_ptr, int32_t* d_coff, int32_t* d_slot2x, int64_t* d_info, void* stream);
int nrm_design_fill(const void* d_x, int x_dtype, int64_t nx, int64_t n, int64_t ldx, int64_t nslots, const int32_t* d_pos, const int32_t* d_w,
					const int64_t* d_base, const int64_t* d_row_ptr, const int32_t* d_coff, int16_t* d_ell, double* d_ellv, int32_t* d_cells,
					double* d_row_vals, int binary, void* stream);
/*
 * single=1's cell selection (association.py:914-918) for a design with entries >= 0, from its CSR form: cell k is selected for grouping i
 * when i's entry is the only one of the cell, and for every grouping when the cell has none.
 *   d_cnt (n) int32 scratch (entries per cell); d_code (n): the cell codes nrm_single1_stream reads; d_seg (nx + 1): grouping i owns the
 *   positions [d_seg[i], d_seg[i + 1]) of d_idx (cell, int64) / d_xe (its value there) / d_ce (position, nc: the covariates there) -- arrays
 *   with room for nnz positions; d_rowinfo (nx, 3): positions of grouping i, smallest and largest value among them (+-inf: none);
 *   d_gram_part (nb (nb + 1) / 2, nrm_single1_select_gram_blocks(), 64), nb = ceil(nc / 8): partial sums of the covariate Gram matrix over
 *   the cells without entries, by 8 x 8 blocks of covariates (block pairs bi <= bj row by row): the caller adds the partials of a pair in order.
 *   d_info (int64[8]): [3] = cells without entries, [4] = positions in all.
 */
int64_t nrm_single1_select_gram_blocks(void);
int nrm_single1_select(const int64_t* d_row_ptr, const int32_t* d_cells, const double* d_vals, int64_t nx, int64_t n, int64_t nnz, const double* d_c,
					   int64_t ldc, int64_t nc, int32_t* d_cnt, int32_t* d_code, int64_t* d_seg, int64_t* d_idx, double* d_xe, double* d_ce,
					   double* d_rowinfo, double* d_gram_part, int64_t* d_info, void* stream);
/* The last step of nrm_single1_select by itself (d_gram_part from d_cnt as the selection left it): a caller that passes d_gram_part = NULL to
 * nrm_single1_select runs it where it likes -- beside nrm_single1_stream, which needs the cell codes only, on another stream. */
int nrm_single1_common_gram(const int32_t* d_cnt, int64_t n, const double* d_c, int64_t ldc, int64_t nc, double* d_gram_part, void* stream);

/*
 * Pseudo-inverses and ranks of a stack of small symmetric matrices (host only): what single=1 needs per grouping (association.py:350-351)
 * and normvar per gene (norm.py:232-246), by the reference's rule (association.py:77-80: singular values below tol x the largest count as
 * zero) -- Jacobi iteration per matrix, the stack dealt to `threads` host threads (0 = choose).  m, inv (count, n, n) fp64; n <= 32.
 */
int nrm_small_pinv(const double* m, int64_t count, int64_t n, double tol, double* inv, int64_t* rank, int threads);

/*
 * Host -> device copy of a caller's (pageable) array through a ring of page-locked staging blocks the library owns: host threads fill
 * the next block while the previous block's DMA runs, so the copy runs at the DMA's rate without page-locking the caller's memory.
 * Returns once the last block has been handed to the DMA engine (completion is in stream order; h_src may be reused at once).
 * threads: host threads per block, 0 = choose.  nrm_upload_release frees the ring.
 */
int nrm_upload(const void* h_src, void* d_dst, int64_t bytes, int threads, void* stream);
int nrm_upload_release(void);

/*
 * Minimum, maximum and number of NaNs of a host array in one threaded pass (host only): what the reference's assertions on its results
 * (de.py:124-131, norm.py:286-289: finite, within range) need.  out[0] = minimum, out[1] = maximum (NaNs left out), out[2] = NaN count.
 */
int nrm_host_minmax(const void* p, int dtype, int64_t count, int threads, double* out);

/*
 * Text matrices of the command line (host only): the reference reads with numpy.loadtxt(delimiter='\t') and writes with
 * numpy.savetxt(fmt='%.8G') (run.py:20-35).  Same text in, same text out, parsed / printed by `threads` host threads (0 = choose).
 *   nrm_tsv_shape: rows = lines with data ('#' comments and blank lines skipped), cols = fields of the first such line.
 *   nrm_tsv_parse: the matrix into out (rows, ld), NRM_F32 or NRM_F64; a field that is not a number, or a row with another number of
 *     fields: NRM_E_ARG (ValueError, as numpy.loadtxt), the place in nrm_last_error().
 *   nrm_tsv_format: rows x cols as text; kind 0 = '%.8G' (NRM_F32 / NRM_F64), kind 1 = '%i' (NRM_TSV_I64 / _I32 / _U8).  Rows are dealt in
 *     order to `parts` threads; part t writes at out + t * part_cap, its length to lens[t];
 *     part_cap >= ceil(rows / parts) * max(cols, 1) * nrm_tsv_width(kind).
 */
#define NRM_TSV_I64 16
#define NRM_TSV_I32 17
#define NRM_TSV_U8 18
int nrm_tsv_shape(const char* buf, int64_t len, int delim, int threads, int64_t* rows, int64_t* cols);
int nrm_tsv_parse(const char* buf, int64_t len, int delim, int threads, void* out, int out_dtype, int64_t rows, int64_t cols, int64_t ld);
int64_t nrm_tsv_width(int kind);
int nrm_tsv_format(const void* data, int dtype, int64_t rows, int64_t cols, int64_t ld, int delim, int kind, char* out, int64_t part_cap,
				   int64_t* lens, int parts);

#ifdef __cplusplus
}
#endif
#endif
