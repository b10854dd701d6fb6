// The host-only numerical helpers of csrc/nrm_small_pinv.hip under g++ -fsanitize=address,undefined (tests/test_cabi_cpu.py builds that file beside this harness):
// the threaded stack of small pseudo-inverses (single=1's groupings, normvar1's genes), the one-pass minimum / maximum / NaN count behind the reference's result
// assertions, the small eigenvalue routine of single=4's rank certificate -- odd counts, every thread count, exact-size heap buffers.
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/normalisr_hip.h"

static char g_err[1024];
void nrm_set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}
#define CHECK(c)                                                                  \
	do {                                                                          \
		if (!(c)) {                                                               \
			fprintf(stderr, "%s:%d: check failed: %s (%s)\n", __FILE__, __LINE__, #c, g_err); \
			exit(1);                                                              \
		}                                                                         \
	} while (0)
static unsigned long long g_st = 0x2545F4914F6CDD1Dull;
static double rnd() {
	g_st ^= g_st << 13;
	g_st ^= g_st >> 7;
	g_st ^= g_st << 17;
	return (double)(g_st >> 11) / 9007199254740992.0 - 0.5;
}

int main() {
	for (int n : {1, 3, 5, 8, 21, 32})
		for (int64_t count : {0, 1, 7, 255, 256, 257, 1500})
			for (int threads : {0, 1, 3, 16}) {
				std::vector<double> m((size_t)(count * n * n)), inv(m.size());
				std::vector<int64_t> rank((size_t)count, -1);
				for (int64_t g = 0; g < count; g++) {  // H D H with D = diag(1 .. r ones, 0 ..) and H a random reflection: rank r = g % n + 1 exactly, eigenvalues 0 / 1
					const int r = (int)(g % n) + 1;
					std::vector<double> v((size_t)n);
					double vv = 0;
					for (auto& x : v) {
						x = rnd();
						vv += x * x;
					}
					for (int i = 0; i < n; i++)
						for (int j = 0; j < n; j++) {
							double acc = 0;
							for (int k = 0; k < r; k++) {
								const double hik = (i == k ? 1.0 : 0.0) - 2.0 * v[(size_t)i] * v[(size_t)k] / vv, hkj = (k == j ? 1.0 : 0.0) - 2.0 * v[(size_t)k] * v[(size_t)j] / vv;
								acc += hik * hkj;
							}
							m[(size_t)(g * n * n + i * n + j)] = acc;
						}
				}
				CHECK(nrm_small_pinv(m.data(), count, n, 1e-8, inv.data(), rank.data(), threads) == NRM_OK);
				for (int64_t g = 0; g < count; g++) {
					if (rank[(size_t)g] != (g % n) + 1) fprintf(stderr, "n %d count %lld threads %d g %lld rank %lld\n", n, (long long)count, threads, (long long)g, (long long)rank[(size_t)g]);
					CHECK(rank[(size_t)g] == (g % n) + 1);
				}
			}
	std::vector<double> bad(33 * 33), out(33 * 33);
	int64_t rk;
	CHECK(nrm_small_pinv(bad.data(), 1, 33, 1e-8, out.data(), &rk, 1) == NRM_E_ARG);  // larger than the stack allows
	// minimum / maximum / NaN count
	for (int64_t count : {0, 1, 5, (1 << 20) - 1, (1 << 20) + 3, 3 * (1 << 20) + 17})
		for (int threads : {0, 1, 5})
			for (int f32 = 0; f32 < 2; f32++) {
				std::vector<double> a((size_t)count);
				std::vector<float> b((size_t)count);
				double mn = INFINITY, mx = -INFINITY, nan = 0;
				for (int64_t i = 0; i < count; i++) {
					double v = rnd() * 100;
					if (i % 100003 == 7) v = NAN;
					b[(size_t)i] = (float)v;
					a[(size_t)i] = f32 ? (double)b[(size_t)i] : v;
					if (v != v) nan += 1;
					else {
						mn = std::fmin(mn, a[(size_t)i]);
						mx = std::fmax(mx, a[(size_t)i]);
					}
				}
				double res[3];
				CHECK(nrm_host_minmax(f32 ? (const void*)b.data() : (const void*)a.data(), f32 ? NRM_F32 : NRM_F64, count, threads, res) == NRM_OK);
				CHECK(res[0] == mn && res[1] == mx && res[2] == nan);
			}
	// eigenvalues
	for (int n : {1, 4, 32}) {
		std::vector<double> m((size_t)n * n, 0.0), w((size_t)n);
		for (int i = 0; i < n; i++) m[(size_t)i * n + i] = (double)(n - i);
		CHECK(nrm_small_eigvals(m.data(), n, w.data()) == NRM_OK);
		for (int i = 0; i < n; i++) CHECK(std::fabs(w[(size_t)i] - (double)(i + 1)) < 1e-12);
	}
	printf("small numerics ok\n");
	return 0;
}
