// Host-side logic of the library that needs no GPU: the tile order and persistent schedule of the Gram kernels (whole tiles ->
// K-aligned parts -> stream-K units), the device scratch pool of the whole-problem entry, the P-value plan.  Kept free of HIP
// headers so that g++ can build it with -fsanitize=address,undefined for the CPU test-suite (tests/host_sanitize.cpp); the same
// functions are compiled for the device where the kernels need them (NRM_HD).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>
#include "../../include/normalisr_hip.h"

#if defined(__HIPCC__)
#define NRM_HD __host__ __device__
#else
#define NRM_HD
#endif

void nrm_set_error(const char* fmt, ...);
#ifndef NRM_REQUIRE
#define NRM_REQUIRE(cond, ...)      \
	do {                            \
		if (!(cond)) {              \
			nrm_set_error(__VA_ARGS__); \
			return NRM_E_ARG;       \
		}                           \
	} while (0)
#endif

NRM_HD inline int nrm_imin(int a, int b) { return a < b ? a : b; }

#define GM 128
#define GN 128

// Tile order: the tile grid is cut into 8x8 super-blocks that are visited one after another (row-major
// inside a super-block).  64 consecutive tiles -- what the 64 co-resident workgroups of one XCD process at
// the same time -- therefore touch 8 A panels and 8 B panels instead of 1 + 64, and those slabs are shared
// through the XCD's L2 while the workgroups advance through K in lockstep.  Symmetric launches keep only
// super-blocks and tiles on or above the diagonal (association.py:893-894).
#define GSB 8
NRM_HD inline __attribute__((always_inline)) void gram_tile_coords(int t, int symmetric, int ntm, int ntn, int& ti, int& tj) {
	const int nbm = (ntm + GSB - 1) / GSB, nbn = (ntn + GSB - 1) / GSB;
	for (int bi = 0; bi < nbm; bi++) {
		const int h = nrm_imin(GSB, ntm - bi * GSB);
		for (int bj = symmetric ? bi : 0; bj < nbn; bj++) {
			const int w = nrm_imin(GSB, ntn - bj * GSB);
			const bool diag = symmetric && bi == bj;
			const int cnt = diag ? h * (h + 1) / 2 : h * w;
			if (t < cnt) {
				int li, lj;
				if (!diag) {
					li = t / w;
					lj = t - li * w;
				} else {
					li = 0;
					int len = h;
					while (t >= len) {
						t -= len;
						li++;
						len--;
					}
					lj = li + t;
				}
				ti = bi * GSB + li;
				tj = bj * GSB + lj;
				return;
			}
			t -= cnt;
		}
	}
	ti = 0;
	tj = 0;
}

struct GramSched {
	int m_rows, n_rows;  // valid (unpadded) rows of A and B
	int ntm, ntn;   // tile grid (M, N)
	int nkt;        // k-tiles (slabs of GK cells)
	int tiles_dp;   // tiles processed whole, one per workgroup per wave
	int tiles_al;   // tiles cut into `parts` equal K ranges, one range per workgroup (K-aligned: slabs still shared in L2)
	int parts;
	int tiles_sk;   // tiles of the tail, cut into unit ranges
	int units_per_wg;
	int nwg;        // persistent workgroups (multiple of 8)
	int tile0;      // first tile of this launch in the gram_tile_coords order (band launches)
	int accumulate; // add to C instead of overwriting it (cell-chunked launches of the sharded path)
	double* work;   // slabs of partial pieces: [tiles_al*parts] then [2 per workgroup]
};



// The persistent loop of a Gram kernel: calls piece(tile, k0, k1, slab) for every piece of workgroup `block` -- whole tiles
// first (one per workgroup per wave, K-lockstep), then its K-aligned part, then its share of the stream-K tail.
template <typename F>
NRM_HD inline __attribute__((always_inline)) void gram_pieces_of(const GramSched& s, int block, F piece) {
	// workgroups that share an XCD (same block index % 8) take consecutive tiles so that operand panels are shared in its L2
	const int per_xcd = s.nwg >> 3;
	const int p = (block & 7) * per_xcd + (block >> 3);
	int t_dp = p;
	bool al_todo = p < s.tiles_al * s.parts;
	int64_t u = (int64_t)p * s.units_per_wg;
	const int64_t total = (int64_t)s.tiles_sk * s.nkt;
	int64_t uend = u + s.units_per_wg;
	if (uend > total) uend = total;
	int sk_piece = 0;
	for (;;) {
		int t, k0, k1;
		double* slab = nullptr;
		if (t_dp < s.tiles_dp) {
			t = t_dp;
			k0 = 0;
			k1 = s.nkt;
			t_dp += s.nwg;
		} else if (al_todo) {
			al_todo = false;
			const int ta = p / s.parts, part = p - ta * s.parts;
			t = s.tiles_dp + ta;
			k0 = (int)((int64_t)s.nkt * part / s.parts);
			k1 = (int)((int64_t)s.nkt * (part + 1) / s.parts);
			slab = s.work + (int64_t)p * (GM * GN);
		} else if (u < uend) {
			const int ts = (int)(u / s.nkt);
			k0 = (int)(u - (int64_t)ts * s.nkt);
			int64_t k1l = k0 + (uend - u);
			k1 = k1l > s.nkt ? s.nkt : (int)k1l;
			t = s.tiles_dp + s.tiles_al + ts;
			u += k1 - k0;
			if (!(k0 == 0 && k1 == s.nkt)) slab = s.work + ((int64_t)s.tiles_al * s.parts + 2 * p + sk_piece) * (GM * GN);
			sk_piece++;
		} else {
			break;
		}
		piece(t, k0, k1, slab);
	}
}

// Doubles of scratch a launch on nwg persistent workgroups can touch: at most nwg slabs of K-aligned parts (tiles_al * parts <= nwg)
// plus two stream-K slabs per workgroup.
static inline int64_t nrm_host_gram_workspace_doubles(int nwg) { return (int64_t)3 * nwg * GM * GN; }

// Tiles that precede super-block row `bi` in the gram_tile_coords order.
static inline int64_t gram_tiles_before(int64_t bi, int symmetric, int64_t ntm, int64_t ntn) {
	const int64_t nbn = (ntn + GSB - 1) / GSB;
	int64_t t = 0;
	for (int64_t b = 0; b < bi; b++) {
		const int64_t h = std::min<int64_t>(GSB, ntm - b * GSB);
		for (int64_t bj = symmetric ? b : 0; bj < nbn; bj++) {
			const int64_t w = std::min<int64_t>(GSB, ntn - bj * GSB);
			t += (symmetric && b == bj) ? h * (h + 1) / 2 : h * w;
		}
	}
	return t;
}


// Host side: the schedule of one launch over output rows [row0, row1) with nkt K-units per tile on nwg persistent workgroups.
static inline int gram_plan(GramSched& s, int64_t m_pad, int64_t n_pad, int64_t nkt, int symmetric, int64_t m_rows, int64_t n_rows, int64_t row0,
							int64_t row1, int nwg, double* work) {
	const int64_t ntm = m_pad / GM, ntn = n_pad / GN;
	NRM_REQUIRE((symmetric ? ntn * (ntn + 1) / 2 : ntm * ntn) < (1LL << 30), "nrm_gram: problem too large for one launch");
	const int64_t tile0 = gram_tiles_before(row0 / (GSB * GM), symmetric, ntm, ntn);
	const int64_t tiles = gram_tiles_before((row1 + GSB * GM - 1) / (GSB * GM), symmetric, ntm, ntn) - tile0;
	NRM_REQUIRE(tiles < (1LL << 30) && nkt < (1LL << 30), "nrm_gram: problem too large for one launch");
	s.tile0 = (int)tile0;
	s.accumulate = 0;
	s.m_rows = (int)((m_rows > 0 && m_rows < m_pad) ? m_rows : m_pad);
	s.n_rows = (int)((n_rows > 0 && n_rows < n_pad) ? n_rows : n_pad);
	s.ntm = (int)ntm;
	s.ntn = (int)ntn;
	s.nkt = (int)nkt;
	s.nwg = nwg - nwg % 8;
	// three phases, every workgroup does the same amount of work in each:
	//  1. whole tiles, one per workgroup per wave (K-lockstep, plain stores);
	//  2. of the remaining rem < nwg tiles, nwg/parts tiles are cut into `parts` equal K ranges (still K-aligned within a
	//     part, so workgroups of an XCD keep sharing slabs through L2);
	//  3. the rest is cut into equal unit ranges (stream-K; different K offsets, no sharing -- kept small).
	const int64_t waves = tiles / s.nwg, rem = tiles - waves * s.nwg;
	s.tiles_dp = (int)(waves * s.nwg);
	s.parts = 1;
	s.tiles_al = 0;
	for (int parts = 2; parts <= 8 && s.nkt >= 8 * parts; parts *= 2)
		if (rem >= s.nwg / parts) {
			s.parts = parts;
			s.tiles_al = s.nwg / parts;
			break;
		}
	const int64_t sk = rem - s.tiles_al;
	s.tiles_sk = (int)sk;
	s.units_per_wg = (int)((sk * s.nkt + s.nwg - 1) / s.nwg);
	s.work = work;
	return NRM_OK;
}

// ---- P-value plan (host constants of p = I_{1-R^2}(dof/2, 1/2), see nrm_pvalue.h) ----------------------------------------------
// Taylor coefficients h_k of sqrt((s/2)/sinh(s/2)) = sum_k h_k s^(2k)
static const double kNrmH[] = {1.0,
							-0.02083333333333333333333,
							0.000390625,
							-0.000007879670965608465608466,
							1.696766579172178130511e-7,
							-3.805064191721906565657e-9,
							8.748377596315407304061e-11,
							-2.044523359411973817584e-12,
							4.833351797967704408319e-14,
							-1.152434101767385923873e-15,
							2.766052043599370042286e-17};

// ln( Gamma(a+1/2)/Gamma(a) ): recurrence up to a >= 24, then the asymptotic series (DLMF 5.11.13).
static inline double nrm_ln_gamma_ratio_half(double a) {
	double shift = 0.0;
	while (a < 24.0) {
		shift += std::log(a / (a + 0.5));
		a += 1.0;
	}
	double i = 1.0 / a, i2 = i * i;
	double s = i * (-1.0 / 8 + i2 * (1.0 / 192 + i2 * (-1.0 / 640 + i2 * (17.0 / 14336 + i2 * (-31.0 / 18432 + i2 * (691.0 / 180224))))));
	return 0.5 * std::log(a) + s + shift;
}

static inline int nrm_pvalue_plan_init_host(nrm_pvalue_plan* plan, double dof) {
	NRM_REQUIRE(plan != nullptr, "nrm_pvalue_plan_init: null plan");
	NRM_REQUIRE(dof > 0, "Insufficient number of cells: dof = %g must be positive", dof);
	const int K = NRM_PCOEF / 2;  // series terms k = 0..K
	double a = 0.5 * dof;
	plan->a = a;
	plan->alpha = a - 0.25;
	plan->ln_front = nrm_ln_gamma_ratio_half(a) - 0.57236494292470008707;  // - ln(pi)/2
	for (int j = 0; j < NRM_PCOEF; j++) plan->coef[j] = 0.0;
	if (a < 8.0) {
		plan->umax = 0.0;  // asymptotic series in 1/alpha not accurate enough: continued fraction only
		return NRM_OK;
	}
	plan->umax = 1.5;
	// Products of half-integers that do not depend on dof, built once: cp[k] = prod_{i<2k} (i + 1/2), pr[j][k] = prod_{i=j+1}^{2k-1} (i + 1/2).
	// (A single=1 screen asks for a plan per grouping: with powl() and these products inside the loops a plan cost 25 us.)
	struct Tab {
		long double cp[NRM_PCOEF / 2 + 1], pr[NRM_PCOEF][NRM_PCOEF / 2 + 1];
		Tab() {
			for (int k = 0; k <= NRM_PCOEF / 2; k++) {
				cp[k] = 1;
				for (int i = 0; i < 2 * k; i++) cp[k] *= (i + 0.5L);
				for (int j = 0; j < NRM_PCOEF; j++) {
					pr[j][k] = 1;
					for (int i = j + 1; i < 2 * k; i++) pr[j][k] *= (i + 0.5L);
				}
			}
		}
	};
	static const Tab tab;
	long double ipow[NRM_PCOEF + 1];  // alpha^-m
	{
		long double pw = 1;
		ipow[0] = 1;
		for (int m = 1; m <= NRM_PCOEF; m++) {
			pw *= (long double)plan->alpha;
			ipow[m] = 1 / pw;
		}
	}
	// S = sum_k h_k alpha^-2k c'_2k,  c'_m = prod_{i<m} (i + 1/2)
	long double S = 0;
	for (int k = 0; k <= K; k++) S += (long double)kNrmH[k] * ipow[2 * k] * tab.cp[k];
	// coef_j = (1/(S sqrt(pi))) sum_{k: 2k > j} h_k alpha^(j-2k) prod_{i=j+1}^{2k-1} (i + 1/2)
	for (int j = 0; j < NRM_PCOEF; j++) {
		long double c = 0;
		for (int k = 1; k <= K; k++) {
			int m = 2 * k;
			if (j >= m) continue;
			c += (long double)kNrmH[k] * ipow[m - j] * tab.pr[j][k];
		}
		plan->coef[j] = (double)(c / (S * 1.7724538509055160272981674833411L));
	}
	return NRM_OK;
}


// ---- scratch pool of the whole-problem host entry -----------------------------------------------------------------------------
// Device scratch of the host entry, kept between calls (hipMalloc / hipFree of GB-sized buffers cost milliseconds
// each): blocks return to a per-process pool and are reused best-fit; nrm_release_cache() frees them.
template <typename Alloc>
struct DevPoolT {
	Alloc mem;  // raw allocator: void* alloc(size_t) (nullptr on failure), void free(void*)
	struct Block {
		void* p;
		size_t cap;
		bool used;
	};
	std::mutex mu;
	std::vector<Block> blocks;
	void* take(size_t bytes) {
		std::lock_guard<std::mutex> g(mu);
		int best = -1;
		for (size_t i = 0; i < blocks.size(); i++)
			if (!blocks[i].used && blocks[i].cap >= bytes && (best < 0 || blocks[i].cap < blocks[(size_t)best].cap)) best = (int)i;
		if (best >= 0 && blocks[(size_t)best].cap <= 2 * bytes + (1 << 20)) {
			blocks[(size_t)best].used = true;
			return blocks[(size_t)best].p;
		}
		void* p = mem.alloc(bytes);
		if (!p) {  // out of memory: drop the idle blocks and retry once
			for (size_t i = 0; i < blocks.size();) {
				if (!blocks[i].used) {
					mem.free(blocks[i].p);
					blocks.erase(blocks.begin() + (long)i);
				} else {
					i++;
				}
			}
			p = mem.alloc(bytes);
			if (!p) return nullptr;
		}
		blocks.push_back({p, bytes, true});
		return p;
	}
	void give(void* p) {
		std::lock_guard<std::mutex> g(mu);
		for (auto& b : blocks)
			if (b.p == p) b.used = false;
	}
	void release() {
		std::lock_guard<std::mutex> g(mu);
		for (size_t i = 0; i < blocks.size();) {
			if (!blocks[i].used) {
				mem.free(blocks[i].p);
				blocks.erase(blocks.begin() + (long)i);
			} else {
				i++;
			}
		}
	}
};
