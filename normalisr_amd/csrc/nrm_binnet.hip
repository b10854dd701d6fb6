// binnet: binarise a co-expression P-value matrix at a per-row Benjamini-Hochberg q-value cutoff
// (reference binnet.py:134-173, bh at :77-131) -- the consumer of the coex p-matrix, kept on the device so
// that a 30k x 30k p-matrix never has to cross PCIe.  HBM-bound: the matrix is read from HBM once (a row is held
// in registers -- or stays in L2 for very wide rows -- for the handful of counting passes), one byte per entry is written.
//
// No sort.  For row i with m = ng-1 off-diagonal entries the reference computes, for each distinct value v with
// rank c_v = #{p <= v}:  q_v = v / (c_v / m)  (arithmetic in the matrix dtype), takes the running minimum from
// the top, and keeps entries with q <= qcut.  That is exactly { p <= tau* },  tau* = max{ v : q_v <= qcut }.
// tau* is found with counting passes: k <- #{p <= qcut (1+d) k / m} started from above converges to the
// largest k for which ANY element of rank > k fails the test even with a relative slack d >> rounding error,
// so only the few distinct values just below that bound need the reference's exact floating-point test.
#include "nrm_common.h"
#include <cstdlib>

#define BN_BINS 1024  // bins of the histogram that tells the threshold search where to start

// Workgroups of BS threads (NW = BS / 64 waves); sm holds 3 NW doubles: [0, NW) for the two-barrier reductions below, [NW, 3 NW) the two
// alternating slot sets of the one-barrier counting passes.
template <int NW>
__device__ __forceinline__ double bn_block_sum(double v, double* sm) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
	__syncthreads();
	double t = 0.0;
#pragma unroll
	for (int w = 0; w < NW; w++) t += sm[w];
	return t;
}
template <int NW>
__device__ __forceinline__ double bn_block_max(double v, double* sm) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
	__syncthreads();
	double t = sm[0];
#pragma unroll
	for (int w = 1; w < NW; w++) t = fmax(t, sm[w]);
	return t;
}

// largest value of the matrix dtype that is <= x: (double)v <= x  <=>  v <= bn_floor_to<T>(x) for every v of that dtype
template <typename T>
__device__ __forceinline__ T bn_floor_to(double x);
template <>
__device__ __forceinline__ double bn_floor_to<double>(double x) {
	return x;
}
template <>
__device__ __forceinline__ float bn_floor_to<float>(double x) {
	if (x >= 3.0e38) return 3.0e38f;
	if (x <= -3.0e38) return -3.0e38f;
	float f = (float)x;
	if ((double)f > x) f = nextafterf(f, -INFINITY);
	return f;
}

// Row access policies for the counting passes: GlobalRow re-reads the row (L2-resident) on every pass; RegRow<ITEMS> loads it
// once into registers (ITEMS values per lane, 256 lanes) so that the ~15 passes of a row are register-only.
template <typename T, int BS = 256>
struct GlobalRow {
	static constexpr int THREADS = BS, MASK_BYTES = 16;
	const T* p;
	int64_t ng, self;
	int phase = 0;
	double bad;  // entries outside [0,1] or not finite seen by this lane (binnet.py:151-152)
	__device__ __forceinline__ GlobalRow(const T* row, int64_t ng_, int64_t self_) : p(row), ng(ng_), self(self_), bad(0) {
		for (int64_t j = threadIdx.x; j < ng; j += BS) {
			const double v = (double)p[j];
			if (!(v >= 0.0 && v <= 1.0)) bad += 1.0;
		}
	}
	template <typename F>
	__device__ __forceinline__ void each(F f) const {
		for (int64_t j = threadIdx.x; j < ng; j += BS)
			if (j != self) f((double)p[j]);
	}
	__device__ __forceinline__ int64_t count_le(double x, double* sm) const {
		double c = 0;
		each([&](double pj) {
			if (pj <= x) c += 1.0;
		});
		return (int64_t)bn_block_sum<BS / 64>(c, sm);
	}
	__device__ __forceinline__ double max_le(double x, double* sm) const {
		double m = -1.0;
		each([&](double pj) {
			if (pj <= x) m = fmax(m, pj);
		});
		return bn_block_max<BS / 64>(m, sm);
	}
	__device__ __forceinline__ void histogram(double top, unsigned* hist, int nb) const {
		const double scale = (double)nb / top;
		each([&](double pj) {
			if (pj <= top) {
				int b = (int)(pj * scale);
				atomicAdd(&hist[b < nb ? b : nb - 1], 1u);
			}
		});
	}
	__device__ __forceinline__ double emit(unsigned char* o, double tau, unsigned char*) const {
		double cnt = 0;
		for (int64_t j = threadIdx.x; j < ng; j += BS) {
			const unsigned char b = (j != self && (double)p[j] <= tau) ? 1 : 0;
			o[j] = b;
			cnt += b;
		}
		return cnt;
	}
};
template <typename T, int ITEMS, int BS = 256>
struct RegRow {
	static constexpr int THREADS = BS, NW = BS / 64;
	static constexpr int MASK_BYTES = ITEMS * BS <= 32768 ? ITEMS * BS : 16;  // the row's mask staged in LDS for 16-byte stores (wide rows: direct)
	T v[ITEMS];  // entries outside the row or on the diagonal hold 2 (> any p-value: never counted, never a maximum <= x)
	int64_t ng;
	double bad;
	int phase = 0;
	// A lane owns chunks of VEC consecutive entries (16 bytes: 4 fp32 / 2 fp64): item q is entry elem(q).  The row is read from memory
	// exactly once, with all loads of a lane in flight together and 16 bytes per lane and instruction when the row is 16-byte aligned
	// (8-byte accesses run at 0.54 - 0.70 of that rate, one-byte mask stores far below the packed ones: MI355X_MICROARCH.md)
	static constexpr int VEC = 16 / sizeof(T);
	static_assert(ITEMS % VEC == 0, "whole 16-byte chunks per lane");
	static __device__ __forceinline__ int64_t elem(int q) { return ((int64_t)(q / VEC) * BS + threadIdx.x) * VEC + q % VEC; }
	__device__ __forceinline__ RegRow(const T* row, int64_t ng_, int64_t self) : ng(ng_), bad(0) {
		// (no branch between the loads: a lane's chunks are all requested before the first is looked at.  A chunk that would cross the end
		// of the row is read from the row's last whole chunk instead and masked below)
		if ((uintptr_t)row % 16 == 0 && ng >= VEC) {
			typedef T vt __attribute__((ext_vector_type(VEC)));
			const int64_t last = (ng - VEC) / VEC * VEC;
			vt t[ITEMS / VEC];
#pragma unroll
			for (int c = 0; c < ITEMS / VEC; c++) {
				const int64_t j = elem(c * VEC);
				t[c] = *reinterpret_cast<const vt*>(row + (j + VEC <= ng ? j : last));
			}
#pragma unroll
			for (int c = 0; c < ITEMS / VEC; c++) {
				const int64_t j = elem(c * VEC);
#pragma unroll
				for (int i = 0; i < VEC; i++) v[c * VEC + i] = t[c][i];
				if (j + VEC > ng && j < ng) {  // the one chunk that straddles the end: its leading entries, one by one
#pragma unroll
					for (int i = 0; i < VEC; i++) v[c * VEC + i] = j + i < ng ? row[j + i] : (T)0;
				}
			}
		} else {
#pragma unroll
			for (int q = 0; q < ITEMS; q++) {
				const int64_t j = elem(q);
				v[q] = row[j < ng ? j : 0];
			}
		}
#pragma unroll
		for (int q = 0; q < ITEMS; q++) {
			const int64_t j = elem(q);
			if (!(v[q] >= (T)0 && v[q] <= (T)1)) bad += 1.0;
			if (j >= ng || j == self) v[q] = (T)2;
		}
	}
	template <typename F>
	__device__ __forceinline__ void each(F f) const {
#pragma unroll
		for (int q = 0; q < ITEMS; q++)
			if (v[q] <= (T)1.5) f((double)v[q]);
	}
	// counting in the matrix dtype with wave ballots: one compare per item and no fp64 arithmetic in the ~15 passes of a row
	__device__ __forceinline__ int64_t count_le(double x, double* sm) {
		const T xf = bn_floor_to<T>(fmin(x, 1.5));
		int c = 0;
#pragma unroll
		for (int q = 0; q < ITEMS; q++) c += __popcll(__ballot(v[q] <= xf));
		// one barrier per pass: the four per-wave results alternate between two LDS slots (a slot is rewritten only after
		// the barrier of the pass in between, by which every wave has read it)
		double* slot = sm + NW + NW * (phase & 1);
		phase++;
		if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = (double)c;
		__syncthreads();
		double t = 0.0;
#pragma unroll
		for (int w = 0; w < NW; w++) t += slot[w];
		return (int64_t)t;
	}
	__device__ __forceinline__ double max_le(double x, double* sm) {
		const T xf = bn_floor_to<T>(fmin(x, 1.5));
		T m = (T)-1;
#pragma unroll
		for (int q = 0; q < ITEMS; q++) m = (v[q] <= xf && v[q] > m) ? v[q] : m;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const T w = __shfl_down(m, o, 64);
			m = w > m ? w : m;
		}
		double* slot = sm + NW + NW * (phase & 1);
		phase++;
		if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = (double)m;
		__syncthreads();
		double t = slot[0];
#pragma unroll
		for (int w = 1; w < NW; w++) t = fmax(t, slot[w]);
		return t;
	}
	__device__ __forceinline__ double emit(unsigned char* o, double tau, unsigned char* stage) const {
		double cnt = 0;
		const T tf = bn_floor_to<T>(fmin(tau, 1.5));
		const bool staged = MASK_BYTES == ITEMS * BS && ((uintptr_t)o % 16 == 0);  // the mask leaves in 16-byte pieces from LDS
		const bool vec = ((uintptr_t)o % VEC == 0);
#pragma unroll
		for (int c = 0; c < ITEMS / VEC; c++) {
			const int64_t j = elem(c * VEC);
			unsigned word = 0;
#pragma unroll
			for (int i = 0; i < VEC; i++) {
				const unsigned b = (v[c * VEC + i] <= tf) ? 1u : 0u;  // diagonal and padding hold 2: never selected
				word |= b << (8 * i);
				cnt += b;
			}
			if (staged) {
				if (VEC == 4)
					*reinterpret_cast<unsigned*>(stage + j) = word;
				else
					*reinterpret_cast<unsigned short*>(stage + j) = (unsigned short)word;
			} else if (vec && j + VEC <= ng) {  // the chunk's mask bytes in one store
				if (VEC == 4)
					*reinterpret_cast<unsigned*>(o + j) = word;
				else
					*reinterpret_cast<unsigned short*>(o + j) = (unsigned short)word;
			} else {
#pragma unroll
				for (int i = 0; i < VEC; i++)
					if (j + i < ng) o[j + i] = (unsigned char)((word >> (8 * i)) & 1u);
			}
		}
		if (staged) {
			__syncthreads();
			for (int64_t j = (int64_t)threadIdx.x * 16; j < ng; j += (int64_t)BS * 16) {
				if (j + 16 <= ng)
					*reinterpret_cast<uint4*>(o + j) = *reinterpret_cast<const uint4*>(stage + j);
				else
					for (int64_t e = j; e < ng; e++) o[e] = stage[e];
			}
		}
		return cnt;
	}
	// the entries <= top into a histogram of nb bins over [0, top) in LDS (hist zeroed by the caller)
	__device__ __forceinline__ void histogram(double top, unsigned* hist, int nb) const {
		const T tf = bn_floor_to<T>(fmin(top, 1.5));
		const double scale = (double)nb / top;
#pragma unroll
		for (int q = 0; q < ITEMS; q++)
			if (v[q] <= tf) {
				int b = (int)((double)v[q] * scale);
				atomicAdd(&hist[b < nb ? b : nb - 1], 1u);
			}
	}
};

template <typename T, typename Row>
__global__ void __launch_bounds__(Row::THREADS) k_binnet_rows(const T* __restrict__ p, int64_t ng, int64_t ldp, double qcut, unsigned char* __restrict__ out,
													 int64_t ldo, unsigned long long* __restrict__ total, int32_t* __restrict__ flags, int64_t row0, long long* __restrict__ dbg) {
	constexpr int NW = Row::THREADS / 64;
	__shared__ double sm[3 * NW];  // [0, NW): block reductions with two barriers; [NW, 3 NW): the alternating slots of the counting passes
	const int64_t i = blockIdx.x;
	const T* prow = p + i * ldp;
	const double m = (double)(ng - 1);
	const T qc = (T)qcut;  // the reference compares in the matrix dtype (numpy weak-scalar promotion)
	const double slack = sizeof(T) == 4 ? 1e-5 : 1e-12;
	if (dbg && threadIdx.x == 0) dbg[i * 6] = wall_clock64();
	Row r(prow, ng, row0 + i);  // row i of this block is gene row0 + i: its diagonal entry sits in that column
	// validity (binnet.py:151-152): finite and inside [0,1]
	if (bn_block_sum<NW>(r.bad, sm) > 0 && threadIdx.x == 0) atomicAdd(&flags[0], 1);
	if (dbg && threadIdx.x == 0) dbg[i * 6 + 1] = wall_clock64();
	double x = 2.0;       // every entry is a candidate
	double tau = -1.0;    // tau*: nothing selected yet
	bool none = false;
	{
		// Where to start.  The loop below finds the largest k with k = g(k), g(k) = #{p <= top k / m}, top = qcut (1 + slack), by
		// iterating from above -- a dozen passes for null P-values, a hundred for a dense network.  A histogram of the entries below
		// top (BN_BINS bins, one pass, LDS atomics) bounds that fixed point from above: for k in bin b, [m b / B, m (b+1) / B), g(k) is at
		// most the entries below the bin's upper edge, so bins whose cumulative count (two bins of rounding slack included) stays below
		// m b / B cannot hold it.  The loop then starts at the upper edge of the last bin that can: two or three passes.
		__shared__ unsigned s_hist[BN_BINS];
		__shared__ unsigned s_scan[NW];
		constexpr int PB = (BN_BINS + Row::THREADS - 1) / Row::THREADS;  // bins per thread
		const double top = qcut * (1.0 + slack);
		for (int b = threadIdx.x; b < BN_BINS; b += Row::THREADS) s_hist[b] = 0u;
		__syncthreads();
		r.histogram(top, s_hist, BN_BINS);
		__syncthreads();
		unsigned own[PB], run = 0;
#pragma unroll
		for (int e = 0; e < PB; e++) {
			const int b = threadIdx.x * PB + e;
			own[e] = b < BN_BINS ? s_hist[b] : 0u;
			run += own[e];
		}
		unsigned inc = run;  // inclusive scan of the threads' sums: inside the wave, then over the waves
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const unsigned t = __shfl_up(inc, o, 64);
			if ((int)(threadIdx.x & 63) >= o) inc += t;
		}
		if ((threadIdx.x & 63) == 63) s_scan[threadIdx.x >> 6] = inc;
		__syncthreads();
		unsigned before = 0;
		for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += s_scan[w];
		unsigned cum = before + inc - run;
#pragma unroll
		for (int e = 0; e < PB; e++) {
			const int b = threadIdx.x * PB + e;
			cum += own[e];
			if (b < BN_BINS) s_hist[b] = cum;  // (each thread rewrites only its own bins)
		}
		__syncthreads();
		double best = -1.0;
#pragma unroll
		for (int e = 0; e < PB; e++) {
			const int b = threadIdx.x * PB + e;
			if (b < BN_BINS && (double)s_hist[b + 2 < BN_BINS ? b + 2 : BN_BINS - 1] >= m * (double)b / (double)BN_BINS - 0.5) best = (double)b;
		}
		best = bn_block_max<NW>(best, sm);
		if (best < 0.0)
			none = true;  // not a single entry below top
		else
			x = fmin(2.0, top * (best + 1.0) / (double)BN_BINS * (1.0 + 1e-9));
	}
	for (int guard = 0; guard < 1000000 && !none; guard++) {
		// skip everything that fails the test even with slack: largest fixed point of k <- #{p <= min(x, qcut (1+slack) k/m)}
		int64_t k = r.count_le(x, sm);
		while (k > 0) {
			const double bound = fmin(x, qcut * (1.0 + slack) * (double)k / m);
			const int64_t c = r.count_le(bound, sm);
			if (c == k) {
				x = bound;
				break;
			}
			k = c;
		}
		if (k == 0) break;
		const double v = r.max_le(x, sm);
		if (v < 0.0) break;
		const int64_t c = r.count_le(v, sm);
		// the reference's arithmetic, in the matrix dtype: w = c/m, q = v/w, clipped to [0,1]  (binnet.py:121-125)
		const T w = (T)c / (T)(ng - 1);
		T q = (T)v / w;
		if (!isfinite((double)q)) q = (T)1;
		q = q > (T)1 ? (T)1 : (q < (T)0 ? (T)0 : q);
		if (q <= qc) {
			tau = v;
			break;
		}
		x = nextafter(v, -1.0);  // v fails: continue strictly below it
	}
	if (dbg && threadIdx.x == 0) {
		dbg[i * 6 + 2] = wall_clock64();
		dbg[i * 6 + 4] = r.phase;
	}
	__shared__ __attribute__((aligned(16))) unsigned char s_mask[Row::MASK_BYTES];
	double cnt = r.emit(out + i * ldo, tau, s_mask);
	cnt = bn_block_sum<NW>(cnt, sm);
	if (threadIdx.x == 0 && cnt > 0) atomicAdd(total, (unsigned long long)cnt);
	if (dbg && threadIdx.x == 0) dbg[i * 6 + 3] = wall_clock64();
}

// profiling aid (tools/time_binnet.py): 6 int64 per row -- time stamps of the 100 MHz clock (start, row loaded and checked, threshold found, mask written), counting passes, spare
static long long* g_bn_dbg = nullptr;
extern "C" int nrm_binnet_debug_buffer(void* d_stamps) {
	g_bn_dbg = (long long*)d_stamps;
	return NRM_OK;
}

template <typename T>
static void bn_launch(const T* p, int64_t rows, int64_t ng, int64_t ldp, double qcut, unsigned char* out, int64_t ldo, unsigned long long* total,
					  int32_t* flags, int64_t row0, hipStream_t st) {
	dim3 grid((unsigned)rows);
#define BN_GO(...) hipLaunchKernelGGL((k_binnet_rows<T, __VA_ARGS__>), grid, dim3(__VA_ARGS__::THREADS), 0, st, p, ng, ldp, qcut, out, ldo, total, flags, row0, g_bn_dbg)
	// the row lives in the registers of its workgroup for all counting passes, up to 98 304 fp32 / 61 440 fp64 entries -- a 30 000-gene row
	// is read from HBM exactly once; wider rows are re-read (L2) on every pass
	// (workgroup sizes measured on MI355X, tools/time_binnet.py: 256 threads up to 20 480 entries, 512 above -- smaller workgroups have
	// cheaper barriers per counting pass and more rows in flight per CU)
	if (ng <= 8 * 256)
		BN_GO(RegRow<T, 8>);
	else if (ng <= 32 * 256)
		BN_GO(RegRow<T, 32>);
	else if (ng <= 80 * 256 && sizeof(T) == 4)
		BN_GO(RegRow<T, 80>);
	else if (ng <= 40 * 512)
		BN_GO(RegRow<T, 40, 512>);
	else if (ng <= 60 * 512)
		BN_GO(RegRow<T, 60, 512>);
	else if (ng <= 60 * 1024)
		BN_GO(RegRow<T, 60, 1024>);
	else if (ng <= 96 * 1024 && sizeof(T) == 4)
		BN_GO(RegRow<T, 96, 1024>);
	else
		BN_GO(GlobalRow<T>);
#undef BN_GO
}

extern "C" int nrm_binnet_rows(const void* d_p, int p_dtype, int64_t rows, int64_t ng, int64_t ldp, int64_t row0, double qcut,
							   unsigned char* d_out, int64_t ldo, unsigned long long* d_total, int32_t* d_flags, void* stream) {
	NRM_REQUIRE(p_dtype == NRM_F32 || p_dtype == NRM_F64, "nrm_binnet: bad dtype");
	NRM_REQUIRE(ng > 1 && ldp >= ng && ldo >= ng, "Wrong shape of net or namet.");
	NRM_REQUIRE(rows >= 0 && row0 >= 0 && row0 + rows <= ng, "nrm_binnet_rows: rows [row0, row0 + rows) outside the matrix");
	NRM_REQUIRE(qcut > 0 && qcut < 1, "Q-value cutoff must be between 0 and 1.");
	NRM_REQUIRE(d_p && d_out && d_total && d_flags, "nrm_binnet: null pointer");
	hipStream_t st = (hipStream_t)stream;
	NRM_HIP(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), st));
	if (rows == 0) return NRM_OK;
	if (p_dtype == NRM_F64)
		bn_launch<double>((const double*)d_p, rows, ng, ldp, qcut, d_out, ldo, d_total, d_flags, row0, st);
	else
		bn_launch<float>((const float*)d_p, rows, ng, ldp, qcut, d_out, ldo, d_total, d_flags, row0, st);
	return nrm_check_launch("k_binnet_rows");
}

extern "C" int nrm_binnet(const void* d_p, int p_dtype, int64_t ng, int64_t ldp, double qcut, unsigned char* d_out, int64_t ldo,
						  unsigned long long* d_total, int32_t* d_flags, void* stream) {
	return nrm_binnet_rows(d_p, p_dtype, ng, ng, ldp, 0, qcut, d_out, ldo, d_total, d_flags, stream);
}
