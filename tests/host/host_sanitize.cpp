// CPU build of the library's host-only logic (csrc/nrm_host_logic.h) for g++ -fsanitize=address,undefined
// (tests/test_cabi_cpu.py builds and runs it; GPU sanitizers are not available on the pool).  Exercises:
//  * the persistent Gram schedule: for many shapes, every k-unit of every tile of a launch is covered exactly once by the pieces
//    of the workgroups, split tiles get a slab each inside the workspace, band launches tile the whole problem;
//  * the scratch pool of the whole-problem entry (best-fit reuse, out-of-memory retry, release) on a counting allocator;
//  * the P-value plan constants (finite, monotone in dof);
//  * the Jacobi pseudo-inverse the host stack and normvar's device solve share (csrc/nrm_jacobi.h): M M^+ M = M, integer ranks of full-rank, rank-deficient,
//    tiny- and huge-scaled matrices up to the 32 x 32 the stack allows.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include "../../normalisr_amd/csrc/nrm_host_logic.h"
#include "../../normalisr_amd/csrc/nrm_jacobi.h"

static char g_err[512];
void nrm_set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

#define CHECK(c)                                                              \
	do {                                                                      \
		if (!(c)) {                                                           \
			fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #c); \
			exit(1);                                                          \
		}                                                                     \
	} while (0)

static void check_schedule(int64_t m_pad, int64_t n_pad, int64_t nkt, int symmetric, int64_t row0, int64_t row1, int nwg) {
	GramSched s;
	std::vector<double> work(8);  // only addresses are formed from it
	CHECK(gram_plan(s, m_pad, n_pad, nkt, symmetric, 0, 0, row0, row1, nwg, work.data()) == NRM_OK);
	const int tiles = s.tiles_dp + s.tiles_al + s.tiles_sk;
	std::vector<std::vector<int>> cover((size_t)tiles, std::vector<int>((size_t)nkt, 0));
	std::set<const double*> slabs;
	for (int b = 0; b < s.nwg; b++)
		gram_pieces_of(s, b, [&](int t, int k0, int k1, double* slab) {
			CHECK(t >= 0 && t < tiles && k0 >= 0 && k0 < k1 && k1 <= nkt);
			for (int k = k0; k < k1; k++) cover[(size_t)t][(size_t)k]++;
			if (slab) {
				CHECK(slab >= work.data() && (int64_t)(slab - work.data()) + GM * GN <= nrm_host_gram_workspace_doubles(s.nwg));
				CHECK(slabs.insert(slab).second);  // nobody shares a slab
			} else
				CHECK(k0 == 0 && k1 == nkt);  // a piece without a slab is a whole tile
			int ti, tj;
			gram_tile_coords(s.tile0 + t, symmetric, s.ntm, s.ntn, ti, tj);
			CHECK(ti >= 0 && ti < s.ntm && tj >= 0 && tj < s.ntn && (!symmetric || tj >= ti));
			CHECK((int64_t)ti * GM >= row0 / (GSB * GM) * (GSB * GM) && (int64_t)ti * GM < row1);
		});
	for (int t = 0; t < tiles; t++)
		for (int64_t k = 0; k < nkt; k++) CHECK(cover[(size_t)t][(size_t)k] == 1);
}

struct CountingAlloc {
	size_t live = 0, limit = 1 << 20, calls = 0;
	std::map<void*, size_t> blocks;
	void* alloc(size_t bytes) {
		calls++;
		if (live + bytes > limit) return nullptr;
		void* p = malloc(bytes);
		blocks[p] = bytes;
		live += bytes;
		return p;
	}
	void free(void* p) {
		live -= blocks.at(p);
		blocks.erase(p);
		::free(p);
	}
};

int main() {
	// schedules: shapes of the BASELINE configs (tiles x k-steps), small ones, ragged bands, symmetric and not
	const int64_t shapes[][3] = {{5120, 5120, 313}, {1024, 15104, 1563}, {3840, 3840, 15625}, {128, 128, 3}, {256, 1152, 64}, {2560, 2560, 72},
								 {8192, 8192, 9}, {30080, 30080, 40}, {128, 256, 1}};
	for (auto& sh : shapes)
		for (int sym = 0; sym < 2; sym++) {
			if (sym && sh[0] != sh[1]) continue;
			check_schedule(sh[0], sh[1], sh[2], sym, 0, sh[0], 256);
			check_schedule(sh[0], sh[1], sh[2], sym, 0, sh[0], 304);
			int64_t tiles_all = 0;
			for (int64_t a = 0; a < sh[0]; a += GSB * GM) {  // band launches tile the problem
				const int64_t b = std::min<int64_t>(sh[0], a + GSB * GM);
				check_schedule(sh[0], sh[1], sh[2], sym, a, b, 256);
				tiles_all += gram_tiles_before((b + GSB * GM - 1) / (GSB * GM), sym, sh[0] / GM, sh[1] / GN) - gram_tiles_before(a / (GSB * GM), sym, sh[0] / GM, sh[1] / GN);
			}
			const int64_t ntm = sh[0] / GM, ntn = sh[1] / GN;
			CHECK(tiles_all == (sym ? ntm * (ntm + 1) / 2 : ntm * ntn));
		}
	GramSched bad;
	CHECK(gram_plan(bad, (int64_t)GM << 20, (int64_t)GN << 20, 10, 0, 0, 0, 0, (int64_t)GM << 20, 256, nullptr) == NRM_E_ARG);  // too large for one launch

	// scratch pool
	{
		DevPoolT<CountingAlloc> pool;
		void* a = pool.take(1000);
		void* b = pool.take(5000);
		CHECK(a && b && pool.mem.calls == 2);
		pool.give(a);
		CHECK(pool.take(900) == a && pool.mem.calls == 2);  // best fit: reused
		pool.give(a);
		void* c = pool.take(100);  // too small a request for the 1000-byte block's slack rule (cap <= 2 * bytes + 1 MB holds: reused)
		CHECK(c == a);
		pool.give(c);
		pool.give(b);
		pool.mem.limit = pool.mem.live + 100;  // the next allocation fails until idle blocks are dropped
		void* d = pool.take(3000000);
		CHECK(d == nullptr);
		pool.mem.limit = 1 << 23;
		d = pool.take(3000000);
		CHECK(d != nullptr);
		pool.give(d);
		pool.release();
		CHECK(pool.mem.live == 0 && pool.blocks.empty());
	}

	// P-value plans
	double prev = -1e300;
	for (double dof : {1.0, 2.0, 15.0, 16.0, 297.0, 1e4, 5e5, 4e6}) {
		nrm_pvalue_plan pl;
		CHECK(nrm_pvalue_plan_init_host(&pl, dof) == NRM_OK);
		CHECK(std::isfinite(pl.ln_front) && pl.ln_front > prev && pl.a == 0.5 * dof);
		prev = pl.ln_front;
		for (int j = 0; j < NRM_PCOEF; j++) CHECK(std::isfinite(pl.coef[j]));
		CHECK((pl.umax > 0) == (dof >= 16));
	}
	nrm_pvalue_plan pl;
	CHECK(nrm_pvalue_plan_init_host(&pl, 0.0) == NRM_E_ARG && nrm_pvalue_plan_init_host(nullptr, 3.0) == NRM_E_ARG);
	// the shared Jacobi pseudo-inverse: Moore-Penrose identity and ranks (association.py:77-80)
	{
		unsigned long long st = 88172645463325252ull;
		auto rnd = [&]() {
			st ^= st << 13;
			st ^= st >> 7;
			st ^= st << 17;
			return (double)(st >> 11) / 9007199254740992.0 - 0.5;
		};
		for (int n : {1, 2, 5, 8, 17, 32})
			for (int deficient = 0; deficient < 3; deficient++)
				for (double scale : {1.0, 1e-150, 1e150}) {
					const int r = deficient == 0 ? n : (deficient == 1 ? (n + 1) / 2 : 1);  // rank of the factor
					std::vector<double> f((size_t)n * r), m((size_t)n * n), inv((size_t)n * n), t((size_t)n * n);
					for (auto& x : f) x = rnd();
					for (int i = 0; i < n; i++)
						for (int j = 0; j < n; j++) {
							double acc = 0;
							for (int k = 0; k < r; k++) acc += f[(size_t)i * r + k] * f[(size_t)j * r + k];
							m[(size_t)i * n + j] = acc * scale;
						}
					int64_t rank = -1;
					nrm_small_pinv_one<32>(m.data(), n, 1e-8, inv.data(), &rank);
					CHECK(rank == r);
					double err = 0, nrm = 0;  // M M^+ M = M
					for (int i = 0; i < n; i++)
						for (int j = 0; j < n; j++) {
							double acc = 0;
							for (int k = 0; k < n; k++) acc += m[(size_t)i * n + k] * inv[(size_t)k * n + j];
							t[(size_t)i * n + j] = acc;
						}
					for (int i = 0; i < n; i++)
						for (int j = 0; j < n; j++) {
							double acc = 0;
							for (int k = 0; k < n; k++) acc += t[(size_t)i * n + k] * m[(size_t)k * n + j];
							err = std::max(err, std::fabs(acc - m[(size_t)i * n + j]));
							nrm = std::max(nrm, std::fabs(m[(size_t)i * n + j]));
						}
					CHECK(err <= 1e-9 * nrm);
				}
	}
	printf("host logic ok\n");
	return 0;
}
