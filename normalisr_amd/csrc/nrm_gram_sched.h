// Tile order, persistent schedule (whole tiles -> K-aligned parts -> stream-K units) and the deterministic fix-up of partial
// tiles, shared by the Gram kernels (fp64 matrix cores: nrm_gram.hip; exact int8-sliced: nrm_gram_i8.hip).
#pragma once
#include <algorithm>
#include "nrm_common.h"

#include "nrm_host_logic.h"

typedef double d2_t __attribute__((ext_vector_type(2)));
#ifndef GFIX_ROWS
#define GFIX_ROWS 4  // rows of a split tile per fix-up workgroup (C2: 32 us at 16, 22 at 8, 18 at 4 -- one pass per thread)
#endif

// the persistent loop of this workgroup (see gram_pieces_of)
template <typename F>
__device__ __forceinline__ void gram_for_each_piece(const GramSched& s, F piece) {
	gram_pieces_of(s, (int)blockIdx.x, piece);
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
	// blockIdx.y selects GFIX_ROWS of the tile's 128 rows: enough workgroups to keep loads in flight (a thread's additions are a chain)
	double* ct = C + (int64_t)ti * GM * ldc + (int64_t)tj * GN;
	const int e0 = blockIdx.y * (GFIX_ROWS * GN);
	for (int e = e0 + threadIdx.x * 2; e < e0 + GFIX_ROWS * GN; e += 512) {
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

