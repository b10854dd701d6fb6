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

__device__ __forceinline__ double bn_block_sum(double v, double* sm) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
	__syncthreads();
	return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ double bn_block_max(double v, double* sm) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
	__syncthreads();
	return fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
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
template <typename T>
struct GlobalRow {
	const T* p;
	int64_t ng, self;
	double bad;  // entries outside [0,1] or not finite seen by this lane (binnet.py:151-152)
	__device__ __forceinline__ GlobalRow(const T* row, int64_t ng_, int64_t self_) : p(row), ng(ng_), self(self_), bad(0) {
		for (int64_t j = threadIdx.x; j < ng; j += 256) {
			const double v = (double)p[j];
			if (!(v >= 0.0 && v <= 1.0)) bad += 1.0;
		}
	}
	template <typename F>
	__device__ __forceinline__ void each(F f) const {
		for (int64_t j = threadIdx.x; j < ng; j += 256)
			if (j != self) f((double)p[j]);
	}
	__device__ __forceinline__ int64_t count_le(double x, double* sm) const {
		double c = 0;
		each([&](double pj) {
			if (pj <= x) c += 1.0;
		});
		return (int64_t)bn_block_sum(c, sm);
	}
	__device__ __forceinline__ double max_le(double x, double* sm) const {
		double m = -1.0;
		each([&](double pj) {
			if (pj <= x) m = fmax(m, pj);
		});
		return bn_block_max(m, sm);
	}
	__device__ __forceinline__ double emit(unsigned char* o, double tau) const {
		double cnt = 0;
		for (int64_t j = threadIdx.x; j < ng; j += 256) {
			const unsigned char b = (j != self && (double)p[j] <= tau) ? 1 : 0;
			o[j] = b;
			cnt += b;
		}
		return cnt;
	}
};
template <typename T, int ITEMS>
struct RegRow {
	T v[ITEMS];  // entries outside the row or on the diagonal hold 2 (> any p-value: never counted, never a maximum <= x)
	int64_t ng;
	double bad;
	int phase = 0;
	// the row is read from memory exactly once, with all ITEMS loads of a lane in flight together (a run-time loop of
	// dependent scalar loads made the first version latency-bound: 0.44 TB/s)
	__device__ __forceinline__ RegRow(const T* row, int64_t ng_, int64_t self) : ng(ng_), bad(0) {
#pragma unroll
		for (int q = 0; q < ITEMS; q++) {
			const int64_t j = (int64_t)q * 256 + threadIdx.x;
			v[q] = j < ng ? row[j] : (T)0;
		}
#pragma unroll
		for (int q = 0; q < ITEMS; q++) {
			const int64_t j = (int64_t)q * 256 + threadIdx.x;
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
		double* slot = sm + 4 + 4 * (phase & 1);
		phase++;
		if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = (double)c;
		__syncthreads();
		return (int64_t)(slot[0] + slot[1] + slot[2] + slot[3]);
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
		double* slot = sm + 4 + 4 * (phase & 1);
		phase++;
		if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = (double)m;
		__syncthreads();
		return fmax(fmax(slot[0], slot[1]), fmax(slot[2], slot[3]));
	}
	__device__ __forceinline__ double emit(unsigned char* o, double tau) const {
		double cnt = 0;
		const T tf = bn_floor_to<T>(fmin(tau, 1.5));
#pragma unroll
		for (int q = 0; q < ITEMS; q++) {
			const int64_t j = (int64_t)q * 256 + threadIdx.x;
			const unsigned char b = (v[q] <= tf) ? 1 : 0;  // diagonal and padding hold 2: never selected
			if (j < ng) o[j] = b;
			cnt += b;
		}
		return cnt;
	}
};

template <typename T, typename Row>
__global__ void __launch_bounds__(256) k_binnet_rows(const T* __restrict__ p, int64_t ng, int64_t ldp, double qcut, unsigned char* __restrict__ out,
													 int64_t ldo, unsigned long long* __restrict__ total, int32_t* __restrict__ flags, int64_t row0) {
	__shared__ double sm[12];  // [0,4): block reductions with two barriers; [4,12): the alternating slots of the counting passes
	const int64_t i = blockIdx.x;
	const T* prow = p + i * ldp;
	const double m = (double)(ng - 1);
	const T qc = (T)qcut;  // the reference compares in the matrix dtype (numpy weak-scalar promotion)
	const double slack = sizeof(T) == 4 ? 1e-5 : 1e-12;
	Row r(prow, ng, row0 + i);  // row i of this block is gene row0 + i: its diagonal entry sits in that column
	// validity (binnet.py:151-152): finite and inside [0,1]
	if (bn_block_sum(r.bad, sm) > 0 && threadIdx.x == 0) atomicAdd(&flags[0], 1);
	double x = 2.0;       // every entry is a candidate
	double tau = -1.0;    // tau*: nothing selected yet
	for (int guard = 0; guard < 1000000; guard++) {
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
	double cnt = r.emit(out + i * ldo, tau);
	cnt = bn_block_sum(cnt, sm);
	if (threadIdx.x == 0 && cnt > 0) atomicAdd(total, (unsigned long long)cnt);
}

template <typename T>
static void bn_launch(const T* p, int64_t rows, int64_t ng, int64_t ldp, double qcut, unsigned char* out, int64_t ldo, unsigned long long* total,
					  int32_t* flags, int64_t row0, hipStream_t st) {
	dim3 grid((unsigned)rows);
	if (ng <= 8 * 256)
		hipLaunchKernelGGL((k_binnet_rows<T, RegRow<T, 8>>), grid, dim3(256), 0, st, p, ng, ldp, qcut, out, ldo, total, flags, row0);
	else if (ng <= 32 * 256)
		hipLaunchKernelGGL((k_binnet_rows<T, RegRow<T, 32>>), grid, dim3(256), 0, st, p, ng, ldp, qcut, out, ldo, total, flags, row0);
	else if (ng <= 96 * 256 && sizeof(T) == 4)
		hipLaunchKernelGGL((k_binnet_rows<T, RegRow<T, 96>>), grid, dim3(256), 0, st, p, ng, ldp, qcut, out, ldo, total, flags, row0);
	else
		hipLaunchKernelGGL((k_binnet_rows<T, GlobalRow<T>>), grid, dim3(256), 0, st, p, ng, ldp, qcut, out, ldo, total, flags, row0);
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
