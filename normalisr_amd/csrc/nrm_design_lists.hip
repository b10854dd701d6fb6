// The design matrix of a CRISPR screen (gRNA incidence: BASELINE configs[3] has 1000 x 50 000 entries, 1 % of them set) turned into the
// lists the sparse-design kernels read -- by kernels of this library.  The reference multiplies the dense matrix (association.py:224-235)
// and selects cells from it with dense passes (association.py:914-918); rounds 1-4 here listed its entries with torch.nonzero, bincount,
// cumsum and two argsorts (53 library kernels, 0.9-1.7 ms per new design).  Now:
//
//   nrm_design_count   ONE pass over the matrix: entries per (chunk of DS_CH cells, design row); what the entries are like (all 1? any < 0?)
//   nrm_design_plan    per chunk the design rows are dealt to the positions of k_de_sparse's lanes, sorted by their entries in the chunk
//                      (a counting rank over the <= 1024 rows of a pass, in LDS); widths of the ELL blocks; row offsets of the CSR form;
//                      both prefix sums
//   nrm_design_fill    a second pass over the matrix: a wave per (design row, chunk) writes the row's entries -- cells ascending, ranks
//                      from wave ballots, no atomics: the lists are the same bits run to run -- into the CSR arrays and the ELL blocks,
//                      padding included
//   nrm_single1_select the cell selection of single=1 (association.py:914-918) from the CSR form: entries per cell, the entries that are
//                      alone in their cell grouping by grouping, cell codes, the covariates at those cells, the covariate Gram matrix of the
//                      cells no grouping touches
// Integer work and byte moves, bound by HBM (two reads of the 200 MB matrix) and launch latency; no matrix cores.
#include "nrm_common.h"
#include "nrm_design.h"

namespace {

template <typename T>
struct DlVec {
	static constexpr int V = 16 / (int)sizeof(T);   // values per 16-byte load
	static constexpr int S = DS_CH / (64 * V);      // loads per lane and (row, chunk)
};

// the V values of a lane's group at cell k of a row; cells past n count as zeros.  ALIGNED (16-byte aligned rows, n % V == 0): one load.
template <typename T, bool ALIGNED>
__device__ __forceinline__ void dl_load(const T* __restrict__ row, int64_t k, int64_t n, T (&v)[DlVec<T>::V]) {
	constexpr int V = DlVec<T>::V;
	if constexpr (ALIGNED) {
		typedef T vt __attribute__((ext_vector_type(V)));
		const vt t = *reinterpret_cast<const vt*>(row + (k < n ? k : 0));
#pragma unroll
		for (int j = 0; j < V; j++) v[j] = k < n ? t[j] : (T)0;
	} else {
#pragma unroll
		for (int j = 0; j < V; j++) v[j] = k + j < n ? row[k + j] : (T)0;
	}
}

// ---- pass 1: entries per (chunk, design row) ------------------------------------------------------------------------------------------------
// grid (ceil(nslots / 4), chunks), 256 threads: a wave per (row, chunk) -- 8 KB of the row, eight 16-byte loads per lane in flight.
#define DL_FLAG_SHIFT 16
#define DL_COUNT(v) ((v) & 0xffff)
template <typename T, bool ALIGNED>
__global__ void __launch_bounds__(256) k_dl_count(const T* __restrict__ X, int64_t ldx, int64_t nx, int64_t n, int64_t nslots, int32_t* __restrict__ cnt) {
	constexpr int V = DlVec<T>::V, S = DlVec<T>::S;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t slot = (int64_t)blockIdx.x * 4 + wave;
	const int c = blockIdx.y;
	if (slot >= nslots) return;
	if (slot >= nx) {  // slots past the last design row (the dealing works on whole groups of 64): empty
		if (lane == 0) cnt[(int64_t)c * nslots + slot] = 0;
		return;
	}
	const T* row = X + slot * ldx;
	const int64_t k0 = (int64_t)c * DS_CH;
	T v[S][V];
#pragma unroll
	for (int s = 0; s < S; s++) dl_load<T, ALIGNED>(row, k0 + (int64_t)(s * 64 + lane) * V, n, v[s]);
	int mine = 0;
	unsigned fl = 0;
#pragma unroll
	for (int s = 0; s < S; s++)
#pragma unroll
		for (int j = 0; j < V; j++) {
			const T x = v[s][j];
			if (x != (T)0) {  // (NaN counts as an entry, as it does for torch.nonzero / numpy.nonzero)
				mine++;
				if (x != (T)1) fl |= DL_NOTONE;
				if (x < (T)0) fl |= DL_NEG;
				if (x > (T)1) fl |= DL_GT1;
				if (x == (T)1) fl |= DL_HAS1;
				if (x != x) fl |= DL_NAN;
			}
		}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
	unsigned wfl = 0;
#pragma unroll
	for (int b = 0; b < 5; b++)
		if (__ballot((fl >> b) & 1)) wfl |= 1u << b;
	// the count (<= DS_CH = 2^11) and, above it, what the entries are like: 25 000 waves meeting at one word of flags -- even only to look
	// at it -- took three times as long as the pass over the matrix (103 against 35 us); k_dl_rowsum ORs the bits of a row's chunks together
	if (lane == 0) cnt[(int64_t)c * nslots + slot] = mine | (int32_t)(wfl << DL_FLAG_SHIFT);
}

// ---- the dealing of every chunk ---------------------------------------------------------------------------------------------------------------
// Position p of k_de_sparse's workgroup gathers, in chunk c, for the design row sig[c][p]: inside every block of DS_PASS slots (one pass of that
// kernel) the rows are sorted by their number of entries in the chunk, most first, ties in row order -- the 64 lists a wave walks in step are
// then equally long.  A rank by counting: key = (entries, row) packed, rank = keys that are larger; 1024 broadcast reads of LDS per thread.
// grid (ceil(nslots / DS_PASS), chunks), DS_PASS threads.
__global__ void __launch_bounds__(DS_PASS) k_dl_plan(const int32_t* __restrict__ cnt, int64_t nslots, int ngroups, int32_t* __restrict__ sig, int32_t* __restrict__ pos,
													 int32_t* __restrict__ w) {
	// the key packs (entries of the slot in this chunk) << 10 | (DS_PASS - 1 - t) into one int32, and the ELL offsets are int16: both hold only for these sizes
	static_assert(DS_PASS <= 1024 && DS_CH <= 2048 && DS_CH < 32768, "k_dl_plan: key packing (10 bits of position, 21 of count) and int16 ELL offsets");
	__shared__ alignas(16) int32_t keys[DS_PASS];  // (read back as int4)
	const int c = blockIdx.y, t = threadIdx.x;
	const int64_t lo = (int64_t)blockIdx.x * DS_PASS, slot = lo + t;
	const int my = slot < nslots ? DL_COUNT(cnt[(int64_t)c * nslots + slot]) : -1;
	const int key = my < 0 ? -1 : (my << 10) | (DS_PASS - 1 - t);  // (entries <= DS_CH = 2^11: 21 bits)
	keys[t] = key;
	__syncthreads();
	int rank = 0;
	const int4* k4 = reinterpret_cast<const int4*>(keys);
#pragma unroll 8
	for (int u = 0; u < DS_PASS / 4; u++) {
		const int4 q = k4[u];
		rank += (q.x > key) + (q.y > key) + (q.z > key) + (q.w > key);
	}
	if (my >= 0) {
		sig[(int64_t)c * nslots + lo + rank] = (int32_t)slot;
		pos[(int64_t)c * nslots + slot] = (int32_t)(lo + rank);
		if ((rank & 63) == 0) w[(int64_t)c * ngroups + (lo + rank) / 64] = (my + 7) & ~7;  // the longest list of the 64, in blocks of 8 entries
	}
}

// entries of every design row before each chunk (coff[c][slot]: where the chunk's entries start inside the row's CSR segment), the rows' totals
// (left at row_ptr[slot + 1] for the scan), slot -> design row.  A thread per slot.
__global__ void __launch_bounds__(256) k_dl_rowsum(const int32_t* __restrict__ cnt, int64_t nx, int64_t nslots, int nch, int32_t* __restrict__ coff,
													int64_t* __restrict__ row_ptr, int32_t* __restrict__ slot2x, int64_t* __restrict__ info) {
	const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
	int run = 0;
	unsigned fl = 0;
	if (slot < nslots) {
		for (int c = 0; c < nch; c++) {
			const int32_t v = cnt[(int64_t)c * nslots + slot];
			coff[(int64_t)c * nslots + slot] = run;
			run += DL_COUNT(v);
			fl |= (unsigned)v >> DL_FLAG_SHIFT;
		}
		if (slot < nx) row_ptr[slot + 1] = run;
		if (slot == 0) row_ptr[0] = 0;
		if (slot2x) slot2x[slot] = slot < nx ? (int32_t)slot : -1;
	}
	unsigned wfl = 0;
#pragma unroll
	for (int b = 0; b < 5; b++)
		if (__ballot((fl >> b) & 1)) wfl |= 1u << b;
	if ((threadIdx.x & 63) == 0 && wfl) atomicOr(reinterpret_cast<unsigned*>(info + 2), wfl);  // (a wave per 64 design rows: a handful of atomics)
}

// Prefix sums by ONE workgroup (a thread per contiguous range, the 1024 range sums scanned in LDS): block 0 turns the rows' totals at
// row_ptr[1 ..] into row_ptr (info[0] = all entries); block 1 turns the widths w (blocks of 64 lists) into the ELL offsets base
// (info[1] = padded entries in all).
__device__ __forceinline__ int64_t dl_block_offsets(int64_t mine, int64_t* part, int64_t& total) {
	const int t = threadIdx.x;
	part[t] = mine;
	__syncthreads();
	for (int o = 1; o < 1024; o <<= 1) {
		const int64_t add = t >= o ? part[t - o] : 0;
		__syncthreads();
		part[t] += add;
		__syncthreads();
	}
	total = part[1023];
	return part[t] - mine;
}

__global__ void __launch_bounds__(1024) k_dl_scan(int64_t* __restrict__ row_ptr, int64_t nx, const int32_t* __restrict__ w, int64_t nw, int64_t* __restrict__ base,
												   int64_t* __restrict__ info) {
	__shared__ int64_t part[1024];
	const int t = threadIdx.x;
	if (blockIdx.x == 0) {
		const int64_t per = (nx + 1023) / 1024, a = t * per, b = a + per < nx ? a + per : nx;
		int64_t s = 0;
		for (int64_t i = a; i < b; i++) s += row_ptr[i + 1];
		int64_t total, run = dl_block_offsets(s, part, total);
		for (int64_t i = a; i < b; i++) {
			run += row_ptr[i + 1];
			row_ptr[i + 1] = run;
		}
		if (t == 0) info[0] = total;
	} else if (w) {
		const int64_t per = (nw + 1023) / 1024, a = t * per, b = a + per < nw ? a + per : nw;
		int64_t s = 0;
		for (int64_t i = a; i < b; i++) s += 64 * (int64_t)w[i];
		int64_t total, run = dl_block_offsets(s, part, total);
		for (int64_t i = a; i < b; i++) {
			base[i] = run;
			run += 64 * (int64_t)w[i];
		}
		if (t == 0) info[1] = total;
	}
}

// ---- pass 2: the entries into their places ------------------------------------------------------------------------------------------------
// A wave per (slot, chunk) again.  Entry j of the list (cells ascending: j = entries of earlier loads + entries of lower lanes in this load +
// earlier entries of this lane -- two ballots' worth of bit counting) goes to
//   cells / row_vals [row_ptr[row] + coff[c][row] + j]                                     (CSR: k_design_stats, nrm_single1_select)
//   ell / ellv       [base[c][p / 64] + ((j / 8) * 64 + p % 64) * 8 + j % 8], p = pos[c][row]     (ELL: k_de_sparse)
// and the list's padding up to the width of its block of 64 is written by the same wave (offset DS_CH: the record of zeros).
template <typename T, bool ALIGNED, bool BINARY>
__global__ void __launch_bounds__(256) k_dl_fill(const T* __restrict__ X, int64_t ldx, int64_t nx, int64_t n, int64_t nslots, int ngroups,
												  const int32_t* __restrict__ pos, const int32_t* __restrict__ w, const int64_t* __restrict__ base,
												  const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ coff, int16_t* __restrict__ ell, double* __restrict__ ellv,
												  int32_t* __restrict__ cells, double* __restrict__ row_vals) {
	constexpr int V = DlVec<T>::V, S = DlVec<T>::S;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t slot = (int64_t)blockIdx.x * 4 + wave;
	const int c = blockIdx.y;
	if (slot >= nslots) return;
	int64_t eb = 0;
	int wd = 0, pl = 0;
	if (ell) {
		const int p = pos[(int64_t)c * nslots + slot];
		eb = base[(int64_t)c * ngroups + p / 64];
		wd = w[(int64_t)c * ngroups + p / 64];
		pl = p & 63;
	}
	int r = 0;  // entries of this (row, chunk) so far (the same in every lane)
	if (slot < nx) {
		const T* row = X + slot * ldx;
		const int64_t k0 = (int64_t)c * DS_CH;
		const int64_t e0 = cells ? row_ptr[slot] + coff[(int64_t)c * nslots + slot] : 0;
		T v[S][V];
#pragma unroll
		for (int s = 0; s < S; s++) dl_load<T, ALIGNED>(row, k0 + (int64_t)(s * 64 + lane) * V, n, v[s]);
		const uint64_t below = (1ull << lane) - 1;
#pragma unroll
		for (int s = 0; s < S; s++) {
			uint64_t m[V];
			int before = 0, all = 0;
#pragma unroll
			for (int j = 0; j < V; j++) {
				m[j] = __ballot(v[s][j] != (T)0);
				before += __popcll(m[j] & below);
				all += __popcll(m[j]);
			}
			int at = r + before;
#pragma unroll
			for (int j = 0; j < V; j++)
				if (v[s][j] != (T)0) {
					const int off = (s * 64 + lane) * V + j;
					if (cells) {
						cells[e0 + at] = (int32_t)(k0 + off);
						if constexpr (!BINARY) row_vals[e0 + at] = (double)v[s][j];
					}
					if (ell) {
						const int64_t q = eb + ((int64_t)(at >> 3) * 64 + pl) * 8 + (at & 7);
						ell[q] = (int16_t)off;
						if constexpr (!BINARY) ellv[q] = (double)v[s][j];
					}
					at++;
				}
			r += all;
		}
	}
	if (ell)
		for (int j = r + lane; j < wd; j += 64) {
			const int64_t q = eb + ((int64_t)(j >> 3) * 64 + pl) * 8 + (j & 7);
			ell[q] = (int16_t)DS_CH;
			if constexpr (!BINARY) ellv[q] = 0.0;
		}
}

template <typename T>
bool dl_aligned(const void* d_x, int64_t ldx, int64_t n) {
	return ((uintptr_t)d_x % 16 == 0) && (ldx * sizeof(T)) % 16 == 0 && n % DlVec<T>::V == 0;
}

}  // namespace

extern "C" int nrm_design_count(const void* d_x, int x_dtype, int64_t nx, int64_t n, int64_t ldx, int32_t* d_cnt, int64_t nslots, int64_t* d_info, void* stream) {
	NRM_REQUIRE(d_x && d_cnt && d_info && nx > 0 && n > 0 && ldx >= n && nslots >= nx && nslots % 64 == 0, "nrm_design_count: bad arguments");
	NRM_REQUIRE(x_dtype == NRM_F32 || x_dtype == NRM_F64, "nrm_design_count: bad dtype");
	const int64_t nch = (n + DS_CH - 1) / DS_CH;
	NRM_REQUIRE(nch <= 65535 && n < (1ll << 31) && nslots < (1ll << 31), "nrm_design_count: matrix too large");
	hipStream_t st = (hipStream_t)stream;
	NRM_HIP(hipMemsetAsync(d_info, 0, 8 * sizeof(int64_t), st));
	const dim3 grid((unsigned)((nslots + 3) / 4), (unsigned)nch);
	if (x_dtype == NRM_F64) {
		if (dl_aligned<double>(d_x, ldx, n))
			hipLaunchKernelGGL((k_dl_count<double, true>), grid, dim3(256), 0, st, (const double*)d_x, ldx, nx, n, nslots, d_cnt);
		else
			hipLaunchKernelGGL((k_dl_count<double, false>), grid, dim3(256), 0, st, (const double*)d_x, ldx, nx, n, nslots, d_cnt);
	} else {
		if (dl_aligned<float>(d_x, ldx, n))
			hipLaunchKernelGGL((k_dl_count<float, true>), grid, dim3(256), 0, st, (const float*)d_x, ldx, nx, n, nslots, d_cnt);
		else
			hipLaunchKernelGGL((k_dl_count<float, false>), grid, dim3(256), 0, st, (const float*)d_x, ldx, nx, n, nslots, d_cnt);
	}
	return nrm_check_launch("k_dl_count");
}

extern "C" int nrm_design_plan(const int32_t* d_cnt, int64_t nx, int64_t n, int64_t nslots, int32_t* d_sig, int32_t* d_pos, int32_t* d_w, int64_t* d_base,
							   int64_t* d_row_ptr, int32_t* d_coff, int32_t* d_slot2x, int64_t* d_info, void* stream) {
	NRM_REQUIRE(d_cnt && d_row_ptr && d_coff && d_info && nx > 0 && n > 0 && nslots >= nx && nslots % 64 == 0, "nrm_design_plan: bad arguments");
	const bool want_ell = d_sig != nullptr;
	NRM_REQUIRE(!want_ell || (d_pos && d_w && d_base), "nrm_design_plan: the ELL form needs d_sig, d_pos, d_w and d_base");
	const int64_t nch = (n + DS_CH - 1) / DS_CH, ngroups = nslots / 64;
	hipStream_t st = (hipStream_t)stream;
	if (want_ell)
		hipLaunchKernelGGL(k_dl_plan, dim3((unsigned)((nslots + DS_PASS - 1) / DS_PASS), (unsigned)nch), dim3(DS_PASS), 0, st, d_cnt, nslots, (int)ngroups, d_sig, d_pos, d_w);
	hipLaunchKernelGGL(k_dl_rowsum, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st, d_cnt, nx, nslots, (int)nch, d_coff, d_row_ptr, d_slot2x, d_info);
	hipLaunchKernelGGL(k_dl_scan, dim3(want_ell ? 2 : 1), dim3(1024), 0, st, d_row_ptr, nx, want_ell ? d_w : nullptr, nch * ngroups, d_base, d_info);
	return nrm_check_launch("k_dl_plan");
}

extern "C" int nrm_design_fill(const void* d_x, int x_dtype, int64_t nx, int64_t n, int64_t ldx, int64_t nslots, const int32_t* d_pos, const int32_t* d_w,
							   const int64_t* d_base, const int64_t* d_row_ptr, const int32_t* d_coff, int16_t* d_ell, double* d_ellv, int32_t* d_cells,
							   double* d_row_vals, int binary, void* stream) {
	NRM_REQUIRE(d_x && nx > 0 && n > 0 && ldx >= n && nslots >= nx && nslots % 64 == 0 && (d_ell || d_cells), "nrm_design_fill: bad arguments");
	NRM_REQUIRE(x_dtype == NRM_F32 || x_dtype == NRM_F64, "nrm_design_fill: bad dtype");
	NRM_REQUIRE(!d_ell || (d_pos && d_w && d_base && (binary || d_ellv)), "nrm_design_fill: the ELL form needs d_pos, d_w, d_base (and d_ellv for valued entries)");
	NRM_REQUIRE(!d_cells || (d_row_ptr && d_coff && (binary || d_row_vals)), "nrm_design_fill: the CSR form needs d_row_ptr, d_coff (and d_row_vals for valued entries)");
	const int64_t nch = (n + DS_CH - 1) / DS_CH, ngroups = nslots / 64;
	// (without the CSR form only the slots' lists are written; the row offsets are then not read)
	hipStream_t st = (hipStream_t)stream;
	const dim3 grid((unsigned)((nslots + 3) / 4), (unsigned)nch);
#define DL_FILL(T, AL, BIN)                                                                                                                                  \
	hipLaunchKernelGGL((k_dl_fill<T, AL, BIN>), grid, dim3(256), 0, st, (const T*)d_x, ldx, nx, n, nslots, (int)ngroups, d_pos, d_w, d_base, d_row_ptr, d_coff, d_ell, \
					   d_ellv, d_cells, d_row_vals)
	if (x_dtype == NRM_F64) {
		const bool al = dl_aligned<double>(d_x, ldx, n);
		if (binary) {
			if (al) DL_FILL(double, true, true); else DL_FILL(double, false, true);
		} else {
			if (al) DL_FILL(double, true, false); else DL_FILL(double, false, false);
		}
	} else {
		const bool al = dl_aligned<float>(d_x, ldx, n);
		if (binary) {
			if (al) DL_FILL(float, true, true); else DL_FILL(float, false, true);
		} else {
			if (al) DL_FILL(float, true, false); else DL_FILL(float, false, false);
		}
	}
#undef DL_FILL
	return nrm_check_launch("k_dl_fill");
}

// ---- single=1: the cell selection from the CSR form (association.py:914-918) ----------------------------------------------------------------
// For entries >= 0, "cell k carries no OTHER grouping than i" means: i's entry is the only one of cell k (k joins E_i) or the cell has no
// entry at all (k joins N, shared by every grouping).  S1_COMMON / S1_SKIP as in nrm_single1.hip.
namespace {

#define S1_COMMON (-2)
#define S1_SKIP (-1)

// entries per cell (integer atomics: the sums do not depend on the order)
__global__ void __launch_bounds__(256) k_s1_cellcount(const int32_t* __restrict__ cells, int64_t nnz, int32_t* __restrict__ cnt) {
	const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (e < nnz) atomicAdd(&cnt[cells[e]], 1);
}

// A wave per grouping walks its entries.  PLACE = false: how many are alone in their cell, and the smallest / largest value among those
// (rowinfo[i] = {count, min, max}; +-inf when there are none); the count also lands at seg[i + 1] for the scan.  PLACE = true (after the scan
// turned the counts into seg): the kept entries -- cells ascending -- to their positions seg[i] + rank: the cell, the value, the covariates
// there, and the cell's code.
template <bool PLACE>
__global__ void __launch_bounds__(64) k_s1_rows(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ cells, const double* __restrict__ vals,
												 const int32_t* __restrict__ cnt, int64_t* __restrict__ seg, double* __restrict__ rowinfo, int64_t* __restrict__ idx,
												 double* __restrict__ xe, int32_t* __restrict__ code, const double* __restrict__ C, int64_t ldc, int nc, double* __restrict__ ce) {
	const int64_t i = blockIdx.x;
	const int lane = threadIdx.x;
	const int64_t a = row_ptr[i], b = row_ptr[i + 1];
	int64_t kept = 0;
	double lo = INFINITY, hi = -INFINITY;
	const uint64_t below = (1ull << lane) - 1;
	const int64_t out0 = PLACE ? seg[i] : 0;
	for (int64_t e0 = a; e0 < b; e0 += 64) {
		const int64_t e = e0 + lane;
		const bool in = e < b;
		const int32_t k = in ? cells[e] : 0;
		const bool keep = in && cnt[k] == 1;
		const double v = in ? (vals ? vals[e] : 1.0) : 0.0;
		const uint64_t m = __ballot(keep);
		if constexpr (PLACE) {
			if (keep) {
				const int64_t q = out0 + kept + __popcll(m & below);
				idx[q] = k;
				xe[q] = v;
				code[k] = (int32_t)q;
				for (int c = 0; c < nc; c++) ce[q * nc + c] = C[c * ldc + k];
			}
		} else if (keep) {
			lo = fmin(lo, v);
			hi = fmax(hi, v);
		}
		kept += __popcll(m);
	}
	if constexpr (!PLACE) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			lo = fmin(lo, __shfl_down(lo, o, 64));
			hi = fmax(hi, __shfl_down(hi, o, 64));
		}
		if (lane == 0) {
			seg[i + 1] = kept;
			if (i == 0) seg[0] = 0;
			rowinfo[i * 3] = (double)kept;
			rowinfo[i * 3 + 1] = lo;
			rowinfo[i * 3 + 2] = hi;
		}
	}
}

// codes of the cells without a position: S1_COMMON where no grouping has an entry, S1_SKIP where several have; info[3] = cells of the first kind
__global__ void __launch_bounds__(256) k_s1_codes(const int32_t* __restrict__ cnt, int64_t n, int32_t* __restrict__ code, int64_t* __restrict__ info) {
	const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
	const bool common = k < n && cnt[k] == 0;
	if (k < n) code[k] = common ? S1_COMMON : S1_SKIP;
	const int c = __popcll(__ballot(common));
	if ((threadIdx.x & 63) == 0 && c) atomicAdd(reinterpret_cast<unsigned long long*>(info + 3), (unsigned long long)c);
}

// Covariate Gram matrix over the cells no grouping touches, sum_k [cnt_k == 0] C_c[k] C_d[k], as blocks of 8 x 8 covariates: blockIdx.y =
// the block pair (bi <= bj), blockIdx.x = a range of cells; part[(pair * gridDim.x + blockIdx.x) * 64 + 8 i + j]: partial sums the host adds
// up in order (a fixed order: the same bits run to run).
__global__ void __launch_bounds__(256) k_s1_common_gram(const int32_t* __restrict__ cnt, int64_t n, const double* __restrict__ C, int64_t ldc, int nc, double* __restrict__ part) {
	int bi = 0, bj = 0;
	{
		const int nb = (nc + 7) / 8;
		int p = blockIdx.y;
		for (bi = 0; bi < nb; bi++) {
			if (p < nb - bi) {
				bj = bi + p;
				break;
			}
			p -= nb - bi;
		}
	}
	double acc[8][8];
#pragma unroll
	for (int i = 0; i < 8; i++)
#pragma unroll
		for (int j = 0; j < 8; j++) acc[i][j] = 0.0;
	for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
		if (cnt[k] != 0) continue;
		double l[8], r[8];
#pragma unroll
		for (int i = 0; i < 8; i++) {
			l[i] = bi * 8 + i < nc ? C[(int64_t)(bi * 8 + i) * ldc + k] : 0.0;
			r[i] = bj * 8 + i < nc ? C[(int64_t)(bj * 8 + i) * ldc + k] : 0.0;
		}
#pragma unroll
		for (int i = 0; i < 8; i++)
#pragma unroll
			for (int j = 0; j < 8; j++) acc[i][j] = fma(l[i], r[j], acc[i][j]);
	}
	__shared__ double red[4][64];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
	for (int i = 0; i < 8; i++)
#pragma unroll
		for (int j = 0; j < 8; j++) {
			double t = acc[i][j];
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
			if (lane == 0) red[wv][i * 8 + j] = t;
		}
	__syncthreads();
	if (threadIdx.x < 64) part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void __launch_bounds__(1024) k_s1_scan(int64_t* __restrict__ seg, int64_t nx, int64_t* __restrict__ info) {
	__shared__ int64_t part[1024];
	const int t = threadIdx.x;
	const int64_t per = (nx + 1023) / 1024, a = t * per, b = a + per < nx ? a + per : nx;
	int64_t s = 0;
	for (int64_t i = a; i < b; i++) s += seg[i + 1];
	int64_t total, run = dl_block_offsets(s, part, total);
	for (int64_t i = a; i < b; i++) {
		run += seg[i + 1];
		seg[i + 1] = run;
	}
	if (t == 0) info[4] = total;
}

}  // namespace

extern "C" int64_t nrm_single1_select_gram_blocks(void) { return 64; }

extern "C" int nrm_single1_select(const int64_t* d_row_ptr, const int32_t* d_cells, const double* d_vals, int64_t nx, int64_t n, int64_t nnz, const double* d_c,
								  int64_t ldc, int64_t nc, int32_t* d_cnt, int32_t* d_code, int64_t* d_seg, int64_t* d_idx, double* d_xe, double* d_ce,
								  double* d_rowinfo, double* d_gram_part, int64_t* d_info, void* stream) {
	NRM_REQUIRE(d_row_ptr && d_cells && d_cnt && d_code && d_seg && d_idx && d_xe && d_rowinfo && d_info && nx > 0 && n > 0 && nnz >= 0 && nc >= 0 && nc <= 32,
				"nrm_single1_select: bad arguments (at most 32 covariates)");
	NRM_REQUIRE(nc == 0 || (d_c && d_ce && ldc >= n), "nrm_single1_select: covariates missing");  // (d_gram_part == NULL: the caller takes the shared cells' Gram matrix itself, nrm_single1_common_gram)
	hipStream_t st = (hipStream_t)stream;
	NRM_HIP(hipMemsetAsync(d_cnt, 0, (size_t)n * 4, st));
	NRM_HIP(hipMemsetAsync(d_info + 3, 0, 2 * sizeof(int64_t), st));
	if (nnz) hipLaunchKernelGGL(k_s1_cellcount, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, d_cells, nnz, d_cnt);
	hipLaunchKernelGGL(k_s1_rows<false>, dim3((unsigned)nx), dim3(64), 0, st, d_row_ptr, d_cells, d_vals, d_cnt, d_seg, d_rowinfo, nullptr, nullptr, nullptr, nullptr, 0, 0,
					   nullptr);
	hipLaunchKernelGGL(k_s1_scan, dim3(1), dim3(1024), 0, st, d_seg, nx, d_info);
	hipLaunchKernelGGL(k_s1_codes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_cnt, n, d_code, d_info);
	hipLaunchKernelGGL(k_s1_rows<true>, dim3((unsigned)nx), dim3(64), 0, st, d_row_ptr, d_cells, d_vals, d_cnt, d_seg, nullptr, d_idx, d_xe, d_code, d_c, ldc, (int)nc, d_ce);
	if (nc && d_gram_part) {
		const int nb = (int)(nc + 7) / 8;
		hipLaunchKernelGGL(k_s1_common_gram, dim3((unsigned)nrm_single1_select_gram_blocks(), (unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, st, d_cnt, n, d_c, ldc, (int)nc,
						   d_gram_part);
	}
	return nrm_check_launch("nrm_single1_select");
}

// The last step of nrm_single1_select by itself -- the covariate Gram matrix of the cells no grouping touches, from d_cnt as the selection left it -- for a
// caller that runs it beside the stream kernel (which needs the cell codes only) on another stream: nrm_single1_select with d_gram_part == NULL, then this.
extern "C" int nrm_single1_common_gram(const int32_t* d_cnt, int64_t n, const double* d_c, int64_t ldc, int64_t nc, double* d_gram_part, void* stream) {
	NRM_REQUIRE(d_cnt && d_c && d_gram_part && n > 0 && nc > 0 && nc <= 32 && ldc >= n, "nrm_single1_common_gram: bad arguments");
	const int nb = (int)(nc + 7) / 8;
	hipLaunchKernelGGL(k_s1_common_gram, dim3((unsigned)nrm_single1_select_gram_blocks(), (unsigned)(nb * (nb + 1) / 2)), dim3(256), 0, (hipStream_t)stream, d_cnt, n, d_c, ldc,
					   (int)nc, d_gram_part);
	return nrm_check_launch("k_s1_common_gram");
}
