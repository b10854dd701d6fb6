// Tile order, persistent schedule (whole tiles -> K-aligned parts -> stream-K units) and the deterministic fix-up of partial
// tiles, shared by the Gram kernels (fp64 matrix cores: nrm_gram.hip; exact int8-sliced: nrm_gram_i8.hip).
#pragma once
#include <algorithm>
#include "nrm_common.h"

#define GM 128
#define GN 128

typedef double d2_t __attribute__((ext_vector_type(2)));

// Tile order: the tile grid is cut into 8x8 super-blocks that are visited one after another (row-major
// inside a super-block).  64 consecutive tiles -- what the 64 co-resident workgroups of one XCD process at
// the same time -- therefore touch 8 A panels and 8 B panels instead of 1 + 64, and those slabs are shared
// through the XCD's L2 while the workgroups advance through K in lockstep.  Symmetric launches keep only
// super-blocks and tiles on or above the diagonal (association.py:893-894).
#define GSB 8
__device__ __forceinline__ void gram_tile_coords(int t, int symmetric, int ntm, int ntn, int& ti, int& tj) {
	const int nbm = (ntm + GSB - 1) / GSB, nbn = (ntn + GSB - 1) / GSB;
	for (int bi = 0; bi < nbm; bi++) {
		const int h = min(GSB, ntm - bi * GSB);
		for (int bj = symmetric ? bi : 0; bj < nbn; bj++) {
			const int w = min(GSB, ntn - bj * GSB);
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


// The persistent loop of a Gram kernel: calls piece(tile, k0, k1, slab) for every piece of this workgroup -- whole tiles
// first (one per workgroup per wave, K-lockstep), then its K-aligned part, then its share of the stream-K tail.
template <typename F>
__device__ __forceinline__ void gram_for_each_piece(const GramSched& s, F piece) {
	// workgroups that share an XCD (same blockIdx % 8) take consecutive tiles so that operand panels are shared in its L2
	const int per_xcd = s.nwg >> 3;
	const int p = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
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

// Adds the slabs of every split tile in a fixed order and writes the tile of C.  One workgroup per split tile.
template <int TAG>  // (a template so that each translation unit that launches it gets its own instantiation)
__global__ void __launch_bounds__(256) k_gram_fixup(double* __restrict__ C, int64_t ldc, int symmetric, GramSched s) {
	const int b = blockIdx.x;
	int ti, tj;
	int first, count;       // slab range (aligned tiles) or workgroup range (stream-K tiles)
	int sk_first_local = 0;
	const double* base;
	if (b < s.tiles_al) {
		gram_tile_coords(s.tile0 + s.tiles_dp + b, symmetric, s.ntm, s.ntn, ti, tj);
		base = s.work + (int64_t)b * s.parts * (GM * GN);
		first = 0;
		count = s.parts;
	} else {
		const int ts = b - s.tiles_al;
		gram_tile_coords(s.tile0 + s.tiles_dp + s.tiles_al + ts, symmetric, s.ntm, s.ntn, ti, tj);
		const int64_t u0 = (int64_t)ts * s.nkt, u1 = u0 + s.nkt;
		first = (int)(u0 / s.units_per_wg);
		int last = (int)((u1 - 1) / s.units_per_wg);
		if (last > s.nwg - 1) last = s.nwg - 1;
		count = last - first + 1;
		if (count == 1 && (int64_t)first * s.units_per_wg <= u0 && (int64_t)(first + 1) * s.units_per_wg >= u1) return;  // stored whole
		base = s.work + (int64_t)s.tiles_al * s.parts * (GM * GN);
		sk_first_local = ((int64_t)first * s.units_per_wg / s.nkt) == ts ? 0 : 1;
	}
	// blockIdx.y selects 16 of the tile's 128 rows: 8 workgroups per tile keep enough loads in flight
	double* ct = C + (int64_t)ti * GM * ldc + (int64_t)tj * GN;
	const int e0 = blockIdx.y * (16 * GN);
	for (int e = e0 + threadIdx.x * 2; e < e0 + 16 * GN; e += 512) {
		d2_t acc = (d2_t){0.0, 0.0};
		if (b < s.tiles_al) {
			for (int q = 0; q < count; q++) acc += *reinterpret_cast<const d2_t*>(base + (int64_t)q * (GM * GN) + e);
		} else {
			// a workgroup's first stream-K piece lies in tile floor(p U / nkt), a second piece (if any) in the next tile:
			// only the first contributor of this tile can be on its second piece
			const double* src = base + ((int64_t)2 * first + sk_first_local) * (GM * GN) + e;
			acc = *reinterpret_cast<const d2_t*>(src);
			src += (int64_t)(2 - sk_first_local) * (GM * GN);
			for (int q = 1; q < count; q++, src += 2 * (GM * GN)) acc += *reinterpret_cast<const d2_t*>(src);
		}
		d2_t* o = reinterpret_cast<d2_t*>(ct + (int64_t)(e / GN) * ldc + (e % GN));
		*o = s.accumulate ? *o + acc : acc;
	}
}

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
