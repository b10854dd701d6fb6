// Pseudo-inverses and ranks of a STACK of small symmetric matrices (host only, no device code).  single=1 needs one per grouping
// (C_S C_S^T of the grouping's own cells, association.py:350-351), normvar one per gene (norm.py:232-246): a thousand to tens of thousands
// of nc x nc matrices, nc ~ 5.  numpy's stacked SVD calls LAPACK once per matrix under the GIL (3.2 us each: 3.2 ms of the 7.1 ms of a
// single=1 call at BASELINE configs[3] size).  Here: the cyclic Jacobi eigenvalue iteration, which for matrices this small converges in a
// handful of sweeps and delivers every eigenvalue to full relative accuracy; the stack is dealt to host threads.  The rule is the
// reference's (association.py:77-80): singular values (= |eigenvalues| of a symmetric matrix) below tol x the largest count as zero, the
// rank is the number kept, M^+ = V_kept diag(1 / s_kept) V_kept^T.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "nrm_common.h"
#include "nrm_jacobi.h"

#define SP_NMAX 32

namespace {

inline void jacobi(double* a, double* v, double* w, int n) { nrm_jacobi(a, v, w, n); }
inline void one(const double* m, int n, double tol, double* inv, int64_t* rank) { nrm_small_pinv_one<SP_NMAX>(m, n, tol, inv, rank); }

}  // namespace

// m (count, n, n) fp64 symmetric -> inv (count, n, n) = the pseudo-inverses, rank (count) int64.  n <= 32.  threads: 0 = choose.
extern "C" int nrm_small_pinv(const double* m, int64_t count, int64_t n, double tol, double* inv, int64_t* rank, int threads) {
	NRM_REQUIRE(count >= 0 && n >= 1 && n <= SP_NMAX && tol > 0 && (count == 0 || (m && inv && rank)), "nrm_small_pinv: bad arguments (matrices of at most %d x %d)", SP_NMAX, SP_NMAX);
	int t = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
	if (t > 16) t = 16;
	if (t > count / 256 + 1) t = (int)(count / 256 + 1);
	if (t < 1) t = 1;
	auto work = [&](int id) {
		for (int64_t g = id; g < count; g += t) one(m + g * n * n, (int)n, tol, inv + g * n * n, rank + g);
	};
	std::vector<std::thread> th;
	for (int id = 1; id < t; id++) th.emplace_back(work, id);
	work(0);
	for (auto& x : th) x.join();
	return NRM_OK;
}

// ---- minimum, maximum and NaN count of a host array, threaded --------------------------------------------------------------------------
// The reference asserts on every result array that it is finite and within its range (de.py:124-131, norm.py:286-289); with numpy that is
// up to five passes with temporaries (12 ms of an 85 ms de call at BASELINE configs[3] size, 35 ms of a 50 ms normvar call on 400 MB of
// fp64 results).  One pass here, dealt to host threads: the three numbers answer every one of those assertions.
namespace {
template <typename T>
void minmax_range(const T* p, int64_t a, int64_t b, double* out) {
	double mn = INFINITY, mx = -INFINITY;
	int64_t nan = 0;
	for (int64_t i = a; i < b; i++) {
		const double v = (double)p[i];
		nan += v != v;
		mn = v < mn ? v : mn;  // (comparisons with NaN are false: NaNs are counted, not propagated)
		mx = v > mx ? v : mx;
	}
	out[0] = mn;
	out[1] = mx;
	out[2] = (double)nan;
}
}  // namespace

// out[0] = minimum, out[1] = maximum (NaNs left out; +inf / -inf for an array without numbers), out[2] = number of NaNs.
extern "C" int nrm_host_minmax(const void* p, int dtype, int64_t count, int threads, double* out) {
	NRM_REQUIRE(count >= 0 && out && (p || count == 0) && (dtype == NRM_F32 || dtype == NRM_F64), "nrm_host_minmax: bad arguments");
	int t = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
	if (t > 16) t = 16;
	if (t > count / (1 << 20) + 1) t = (int)(count / (1 << 20) + 1);
	if (t < 1) t = 1;
	std::vector<double> part(3 * (size_t)t);
	const int64_t per = (count + t - 1) / t;
	auto work = [&](int id) {
		const int64_t a = per * id < count ? per * id : count, b = a + per < count ? a + per : count;
		if (dtype == NRM_F64)
			minmax_range<double>((const double*)p, a, b, &part[3 * id]);
		else
			minmax_range<float>((const float*)p, a, b, &part[3 * id]);
	};
	std::vector<std::thread> th;
	for (int id = 1; id < t; id++) th.emplace_back(work, id);
	work(0);
	for (auto& x : th) x.join();
	out[0] = INFINITY;
	out[1] = -INFINITY;
	out[2] = 0.0;
	for (int id = 0; id < t; id++) {
		out[0] = part[3 * id] < out[0] ? part[3 * id] : out[0];
		out[1] = part[3 * id + 1] > out[1] ? part[3 * id + 1] : out[1];
		out[2] += part[3 * id + 2];
	}
	return NRM_OK;
}

// eigenvalues (ascending) of one small symmetric matrix (n <= 32): the covariate block of single=4's rank certificate (nrm_host_entries.hip)
extern "C" int nrm_small_eigvals(const double* m, int64_t n, double* w) {
	NRM_REQUIRE(m && w && n > 0 && n <= SP_NMAX, "nrm_small_eigvals: bad arguments (matrices up to %d x %d)", SP_NMAX, SP_NMAX);
	double a[SP_NMAX * SP_NMAX], v[SP_NMAX * SP_NMAX];
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) a[i * n + j] = 0.5 * (m[i * n + j] + m[j * n + i]);
	jacobi(a, v, w, (int)n);
	for (int i = 1; i < n; i++)  // insertion sort: n <= 32
		for (int j = i; j > 0 && w[j] < w[j - 1]; j--) std::swap(w[j], w[j - 1]);
	return NRM_OK;
}
