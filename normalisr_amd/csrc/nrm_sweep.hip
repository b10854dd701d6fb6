// K3: per-pair sweep (R^2 -> p, effect size, Pearson r, t) and the p-value plan.
// Reference lines restated: association.py:231-235 (variance 0->1, gamma, R^2), :248-249 (range
// assertion, beta.cdf), :1037-1057 (coef->dot conversion, symmetrise, zero diagonal).
#include <cmath>
#include <cstring>
#include "nrm_pvalue.h"
#include "nrm_fix.h"
#include "nrm_host_logic.h"

// ---- host: plan -------------------------------------------------------------------------------

extern "C" int nrm_pvalue_plan_init(nrm_pvalue_plan* plan, double dof) { return nrm_pvalue_plan_init_host(plan, dof); }

extern "C" int nrm_pvalue_plan_init_many(const double* dof, int64_t count, double* out, int64_t pitch) {
	NRM_REQUIRE(count >= 0 && pitch >= (int64_t)(sizeof(nrm_pvalue_plan) / sizeof(double)), "nrm_pvalue_plan_init_many: bad sizes");
	NRM_REQUIRE(count == 0 || (dof && out), "nrm_pvalue_plan_init_many: null pointer");
	for (int64_t j = 0; j < count; j++) {
		nrm_pvalue_plan plan;
		int rc = nrm_pvalue_plan_init(&plan, dof[j]);
		if (rc) return rc;
		memcpy(out + j * pitch, &plan, sizeof(plan));
	}
	return NRM_OK;
}

static PvalPlan to_dev(const nrm_pvalue_plan& p) {
	PvalPlan d;
	d.a = p.a;
	d.alpha = p.alpha;
	d.ln_front = p.ln_front;
	d.umax = p.umax;
	for (int j = 0; j < NRM_PCOEF; j++) d.coef[j] = p.coef[j];
	return d;
}

// ---- kernels ----------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_pvalues_from_r2(const double* __restrict__ r2, int64_t count, PvalPlan pl,
														  double* __restrict__ p) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; i < count; i += stride) p[i] = nrm_pvalue(r2[i], pl);
}

#define SW_T 64
#define SW_T_ 64
template <typename T>
__device__ __forceinline__ void store_out(void* base, int64_t idx, double v) {
	reinterpret_cast<T*>(base)[idx] = (T)v;
}

// Integer-engine row records (nrm_fix.h) inside a sweep: the x rows of the workgroup's tile staged in LDS, the thread's column
// The guard's counter saturates instead of wrapping: a wave adds its hits only while the counter is below 2^30 (a plain read first:
// the overshoot is bounded by the waves in flight times the pairs of a wave, far below 2^31), so that "hits > 0" stays true
// however many of a large problem's pairs fail (46 000 sparse genes on the non-symmetric path would pass 2^31), also after the
// ranks' counters have been added up.
__device__ __forceinline__ void nrm_guard_count(int* flags, int bad) {
	if (bad > 0 && __hip_atomic_load(flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (1 << 30)) atomicAdd(flags + 2, bad);
}

// record in registers; fix_finish publishes the guard's verdict: flags[2] += uncertified pairs, flags[3] = max error estimate
// (float bits; positive floats order like their bit patterns).
#define SW_FIX 7  // doubles of a record the sweeps use: u0..u4, c, g
#define PV_FN [](double r2_, const PvalPlan& pl_) { return nrm_pvalue(r2_, pl_); }
__device__ __forceinline__ void fix_stage_rows(const FixArgs& f, double (*sfx)[SW_FIX], int64_t row0, int64_t nrows) {
	if (!f.fx) return;
	for (int i = threadIdx.x; i < SW_T_ * SW_FIX; i += 256) {
		const int r = i / SW_FIX, c = i - r * SW_FIX;
		sfx[r][c] = (row0 + r < nrows) ? f.fx[(row0 + r) * NRM_FIX_STRIDE + c] : 0.0;
	}
}
__device__ __forceinline__ FixCol fix_column(const FixArgs& f, int64_t col, int64_t ncols) {
	FixCol y = {{0.0, 0.0, 0.0, 0.0, 0.0}, 0.0, 0.0};
	if (f.fx && col < ncols) y = nrm_fix_col(f.fy + col * NRM_FIX_STRIDE);
	return y;
}
__device__ __forceinline__ void fix_finish(const FixArgs& f, int32_t* __restrict__ flags, FixAcc& a, const PvalPlan& pl) {
	if (!f.fx || !flags) return;
	nrm_fix_close(a, pl, PV_FN);
	int bad = a.bad;
	float worst = a.worst;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		bad += __shfl_xor(bad, o, 64);
		worst = fmaxf(worst, __shfl_xor(worst, o, 64));
	}
	if ((threadIdx.x & 63) == 0) {
		nrm_guard_count(flags, bad);
		// (a maximum only grows: a wave that cannot raise it stays away -- tens of thousands of atomics on one address cost more than
		// the guard itself; a stale read merely sends a redundant atomic)
		if (worst > 0.f && __float_as_int(worst) > __hip_atomic_load(&flags[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&flags[3], __float_as_int(worst));
	}
}

// One workgroup = one 64x64 tile of the (nx, ny) output.  The source tile of dot is staged through LDS
// so that the mirrored half of a symmetric (coex) problem is read coalesced and transposed on chip.
template <typename OutT>
__global__ void __launch_bounds__(256, 3) k_assoc_sweep(const double* __restrict__ dot, int64_t ldd,
													  const double* __restrict__ ssx, const double* __restrict__ ssy,
													  int64_t nx, int64_t ny, double ncells, double dof, int symmetric,
													  int stat_kind, PvalPlan pl, void* __restrict__ p_out,
													  void* __restrict__ stat_out, void* __restrict__ r_out,
													  void* __restrict__ t_out, int64_t ldo, int32_t* __restrict__ flags, int bi0, FixArgs fix) {
	__shared__ double tile[SW_T][SW_T + 1];
	__shared__ double sx[SW_T], sy[SW_T];
	__shared__ double sfx[SW_T][SW_FIX];
	__shared__ double sfc[SW_T][SW_FIX];  // symmetric problems: the records of the column genes as well (see `swap` below)
	const int bi = blockIdx.y + bi0, bj = blockIdx.x;
	fix_stage_rows(fix, sfx, (int64_t)bi * SW_T, nx);
	if (symmetric && bi >= bj) {
		FixArgs fc = fix;
		fc.fx = fix.fx ? fix.fy : nullptr;
		fix_stage_rows(fc, sfc, (int64_t)bj * SW_T, ny);
	}
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
	// source tile: for coex always from the upper triangle (association.py:1050-1057)
	const bool mirror = symmetric && bi > bj;
	const int si = mirror ? bj : bi, sj = mirror ? bi : bj;
	for (int r = ty; r < SW_T; r += 4) {
		int64_t gi = (int64_t)si * SW_T + r, gj = (int64_t)sj * SW_T + tx;
		int64_t lim_i = mirror ? ny : nx, lim_j = mirror ? nx : ny;
		tile[r][tx] = (gi < lim_i && gj < lim_j) ? dot[gi * ldd + gj] : 0.0;
	}
	if (threadIdx.x < SW_T) {
		int64_t gi = (int64_t)bi * SW_T + threadIdx.x;
		double v = gi < nx ? ssx[gi] : 1.0;
		sx[threadIdx.x] = (v == 0.0) ? ncells : v;  // variance 0 -> 1  (association.py:231)
	} else if (threadIdx.x < 2 * SW_T) {
		int t = threadIdx.x - SW_T;
		int64_t gj = (int64_t)bj * SW_T + t;
		double v = gj < ny ? ssy[gj] : 1.0;
		sy[t] = (v == 0.0) ? ncells : v;  // association.py:233
	}
	__syncthreads();
	int bad_nf = 0, bad_rng = 0;
	FixAcc facc = nrm_fix_acc();
	const double sqrt_dof = sqrt(fix.dof);
	const int64_t gj = (int64_t)bj * SW_T + tx;
	const FixCol fy = fix_column(fix, gj, ny);
	auto pair_dot = [&](int r) {  // x~_i . y~_j of the pair in row r of this thread's column
		double d;
		if (!symmetric)
			d = tile[r][tx];
		else if (mirror)
			d = tile[tx][r];
		else if (bi == bj)
			d = (r <= tx) ? tile[r][tx] : tile[tx][r];
		else
			d = tile[r][tx];
		if (fix.fx) {
			// The correction is symmetric in the pair but not in its rounding: an element below the diagonal of a symmetric problem
			// takes it with the roles of its two genes swapped, i.e. exactly as its mirror image above the diagonal does, so that
			// the results stay bitwise symmetric (association.py:1050-1057 copies the upper triangle)
			const bool swap = symmetric && (mirror || (bi == bj && r > tx));
			d += swap ? nrm_fix_corr(sfc[tx], nrm_fix_col(sfx[r]), fix.top, fix.inv_n) : nrm_fix_corr(sfx[r], fy, fix.top, fix.inv_n);
		}
		return d;
	};
	for (int r = ty; r < SW_T; r += 4) {
		const int64_t gi = (int64_t)bi * SW_T + r;
		if (gi >= nx || gj >= ny) continue;
		const double d = pair_dot(r);
		const double vx = sx[r], vy = sy[tx];
		double r2 = (d * d) / (vx * vy);  // = gamma^2 vx / vy  (association.py:235)
		double p, stat, rr, tt;
		if (symmetric && gi == gj) {
			p = 0.0;  // triu(.,1) + transpose leaves exact zeros on the diagonal (Q1)
			stat = 0.0;
			rr = 0.0;
			tt = 0.0;
		} else {
			if (!isfinite(r2) || !isfinite(vx) || !isfinite(vy)) bad_nf = 1;
			if (r2 > 1.0 + 1e-8) bad_rng = 1;  // association.py:248
			p = nrm_pvalue(r2, pl);
			stat = stat_kind ? d / vx : d / ncells;
			rr = d / sqrt(vx * vy);
			double rc = fmin(r2, 1.0);
			tt = copysign(sqrt(dof * rc / (1.0 - rc)), d);
			if (fix.fx && fix.budget > 0.0) nrm_fix_note(facc, sfx[r][5], sfx[r][6], r2);
		}
		const int64_t o = gi * ldo + gj;
		store_out<OutT>(p_out, o, p);
		store_out<OutT>(stat_out, o, stat);
		if (r_out) store_out<OutT>(r_out, o, rr);
		if (t_out) store_out<OutT>(t_out, o, tt);
	}
	if (fix.fx && fix.budget > 0.0 && nrm_fix_screen(fix, facc, fy, sqrt_dof)) {  // rare: the exact test, pair by pair
#pragma unroll 1
		for (int r = ty; r < SW_T; r += 4) {
			const int64_t gi = (int64_t)bi * SW_T + r;
			if (gi >= nx || gj >= ny || (symmetric && gi == gj)) continue;
			const double d = pair_dot(r), r2 = (d * d) / (sx[r] * sy[tx]);
			nrm_fix_guard(fix, sfx[r][5], sfx[r][6], fy, r2, sqrt_dof, nrm_pvalue(r2, pl) != 0.0, facc);
		}
	}
	if (flags) {
		if (bad_nf) atomicAdd(&flags[0], 1);
		if (bad_rng) atomicAdd(&flags[1], 1);
	}
	fix_finish(fix, flags, facc, pl);
}

// Symmetric (coex) sweep over the upper triangle of 64x64 tiles only: every p-value is computed once and
// written twice (direct and mirrored through an LDS transpose), halving the special-function work.
// A workgroup is one straight line: every thread loads its 16 products straight from memory (coalesced along tx) while the row
// tables are staged, computes, stores its results and leaves a copy in LDS; ONE barrier that waits for LDS only (the stores stay
// in flight: __syncthreads() would drain them, and at 3 workgroups per CU those drains were a third of this kernel's time)
// publishes the tile, and the mirrored block is stored from its transpose.
template <typename OutT, int SR>  // SR: rows of a workgroup's piece of a tile (64, 32 or 16; 32 = two workgroups per tile: a finer tail)
__global__ void __launch_bounds__(256, 3) k_assoc_sweep_sym(const double* __restrict__ dot, int64_t ldd, const double* __restrict__ ss,
														  int64_t ng, int nb, double ncells, PvalPlan pl, OutT* __restrict__ p_out,
														  OutT* __restrict__ stat_out, int64_t ldo, int32_t* __restrict__ flags, int bi0, FixArgs fix) {
	__shared__ OutT tp[SR][SW_T + 1], ts[SR][SW_T + 1];  // results of the piece, read back transposed
	__shared__ double sx[SR], sy[SW_T];
	__shared__ double sfx[SR][SW_FIX];
	constexpr int NI = SR / 4;
	const int h0 = (int)(blockIdx.x % (SW_T / SR)) * SR;  // first row of the piece within its tile
	int b = blockIdx.x / (SW_T / SR), bi = bi0, len = nb - bi0;
	while (b >= len) {
		b -= len;
		bi++;
		len--;
	}
	const int bj = bi + b;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	const int64_t gj = (int64_t)bj * SW_T + tx;
	double dv[NI];
#pragma unroll
	for (int i = 0; i < NI; i++) {
		const int64_t gi = (int64_t)bi * SW_T + h0 + ty + 4 * i;
		dv[i] = (gi < ng && gj < ng) ? dot[gi * ldd + gj] : 0.0;
	}
	if (fix.fx)
		for (int i = threadIdx.x; i < SR * SW_FIX; i += 256) {
			const int r = i / SW_FIX, c = i - r * SW_FIX;
			const int64_t gi = (int64_t)bi * SW_T + h0 + r;
			sfx[r][c] = gi < ng ? fix.fx[gi * NRM_FIX_STRIDE + c] : 0.0;
		}
	if (threadIdx.x >= 128 && threadIdx.x < 128 + SR) {
		const int t = threadIdx.x - 128;
		int64_t gi = (int64_t)bi * SW_T + h0 + t;
		double v = gi < ng ? ss[gi] : 1.0;
		sx[t] = (v == 0.0) ? ncells : v;
	} else if (threadIdx.x >= SW_T && threadIdx.x < 2 * SW_T) {
		int t = threadIdx.x - SW_T;
		int64_t gc = (int64_t)bj * SW_T + t;
		double v = gc < ng ? ss[gc] : 1.0;
		sy[t] = (v == 0.0) ? ncells : v;
	}
	const FixCol fy = fix_column(fix, gj, ng);
	__syncthreads();
	int bad_nf = 0, bad_rng = 0;
	FixAcc facc = nrm_fix_acc();
	const double sqrt_dof = sqrt(fix.dof);
	unsigned slow = 0;  // pairs the straight-line P-value does not cover (R^2 >= 1/4, small dof, non-finite): redone below
#pragma unroll
	for (int i = 0; i < NI; i++) {
		const int r = ty + 4 * i;
		const int64_t gi = (int64_t)bi * SW_T + h0 + r;
		const bool lower = bi == bj && h0 + r > tx;  // below the diagonal of a diagonal tile: filled in from its mirror image afterwards
		const bool pair = gi < ng && gj < ng && gi != gj && !lower;
		double d = dv[i];
		if (fix.fx) d += nrm_fix_corr(sfx[r], fy, fix.top, fix.inv_n);
		const double vx = sx[r], vy = sy[tx];
		const double r2 = (d * d) / (vx * vy);
		bool ok;
		double p = nrm_pvalue_fast(r2, pl, ok);  // no branch around it: the NI evaluations of a thread interleave
		double st = d / ncells;
		if (pair) {
			if (!isfinite(r2) || !isfinite(vx) || !isfinite(vy)) bad_nf = 1;
			if (r2 > 1.0 + 1e-8) bad_rng = 1;
			if (!ok) slow |= 1u << i;
			if (fix.fx && fix.budget > 0.0) nrm_fix_note(facc, sfx[r][5], sfx[r][6], r2);
		} else {
			p = 0.0;
			st = 0.0;
		}
		tp[r][tx] = (OutT)p;
		ts[r][tx] = (OutT)st;
		if (gi < ng && gj < ng && !lower) {
			p_out[gi * ldo + gj] = (OutT)p;
			stat_out[gi * ldo + gj] = (OutT)st;
		}
	}
	if (slow) {
#pragma unroll 1
		for (int i = 0; i < NI; i++) {  // (one copy of the general P-value code: the product comes from memory again)
			if (!(slow >> i & 1)) continue;
			const int r = ty + 4 * i;
			const int64_t gi = (int64_t)bi * SW_T + h0 + r;
			double d = dot[gi * ldd + gj];
			if (fix.fx) d += nrm_fix_corr(sfx[r], fy, fix.top, fix.inv_n);
			const OutT p = (OutT)nrm_pvalue((d * d) / (sx[r] * sy[tx]), pl);
			tp[r][tx] = p;
			p_out[gi * ldo + gj] = p;  // (after the store above, from the same thread: this one stays)
		}
	}
	if (fix.fx && fix.budget > 0.0 && nrm_fix_screen(fix, facc, fy, sqrt_dof)) {  // rare: the exact test, pair by pair
#pragma unroll
		for (int i = 0; i < NI; i++) {
			const int r = ty + 4 * i;
			const int64_t gi = (int64_t)bi * SW_T + h0 + r;
			if (gi >= ng || gj >= ng || gi == gj || (bi == bj && h0 + r > tx)) continue;
			const double d = dv[i] + nrm_fix_corr(sfx[r], fy, fix.top, fix.inv_n);
			nrm_fix_guard(fix, sfx[r][5], sfx[r][6], fy, (d * d) / (sx[r] * sy[tx]), sqrt_dof, tp[r][tx] != (OutT)0, facc);
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tp / ts are complete; the stores above stay in flight
	{
		// mirrored block: rows of block bj, columns h0 .. h0 + SR of block bi; a wave stores 64 / SR rows of SR results at a time
		const int mc = threadIdx.x & (SR - 1), mr0 = threadIdx.x / SR;
		const int64_t oj = (int64_t)bi * SW_T + h0 + mc;
#pragma unroll
		for (int i = 0; i < NI; i++) {
			const int r = mr0 + (256 / SR) * i;
			const int64_t oi = (int64_t)bj * SW_T + r;
			if (oi < ng && oj < ng && (bi != bj || r > h0 + mc)) {
				p_out[oi * ldo + oj] = tp[mc][r];
				stat_out[oi * ldo + oj] = ts[mc][r];
			}
		}
	}
	if (flags) {
		if (bad_nf) atomicAdd(&flags[0], 1);
		if (bad_rng) atomicAdd(&flags[1], 1);
	}
	fix_finish(fix, flags, facc, pl);
}

// Off-diagonal rectangle of a symmetric (coex) problem: rows [r0, r0 + mx) against columns [c0, c0 + my) with c0 + my <= r0.
// Every p-value is computed once and written twice -- at (r0 + i, c0 + j) and, through an LDS transpose, at (c0 + j, r0 + i) --
// so that a pipelined coex (rows arriving chunk by chunk) can finish and ship both halves of a chunk's pairs at once.
// (fp64 outputs need 195 registers: two waves per SIMD is what the kernel gets, and what it asks for)
template <typename OutT>
__global__ void __launch_bounds__(256, sizeof(OutT) == 8 ? 2 : 3) k_assoc_sweep_mirror(const double* __restrict__ dot, int64_t ldd, const double* __restrict__ ssx,
															 const double* __restrict__ ssy, int64_t mx, int64_t my, double ncells, PvalPlan pl,
															 OutT* __restrict__ p_out, OutT* __restrict__ stat_out, int64_t ldo, int64_t r0,
															 int64_t c0, int32_t* __restrict__ flags, FixArgs fix) {
	__shared__ OutT tp[SW_T][SW_T + 1], ts[SW_T][SW_T + 1];  // results of the tile, read back transposed (as in k_assoc_sweep_sym)
	__shared__ double sx[SW_T], sy[SW_T];
	__shared__ double sfx[SW_T][SW_FIX];
	const int bi = blockIdx.y, bj = blockIdx.x;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	const int64_t gj = (int64_t)bj * SW_T + tx;
	double dv[SW_T / 4];
#pragma unroll
	for (int i = 0; i < SW_T / 4; i++) {
		const int64_t gi = (int64_t)bi * SW_T + ty + 4 * i;
		dv[i] = (gi < mx && gj < my) ? dot[gi * ldd + gj] : 0.0;
	}
	fix_stage_rows(fix, sfx, (int64_t)bi * SW_T, mx);
	if (threadIdx.x < SW_T) {
		const int64_t gi = (int64_t)bi * SW_T + threadIdx.x;
		const double v = gi < mx ? ssx[gi] : 1.0;
		sx[threadIdx.x] = (v == 0.0) ? ncells : v;  // variance 0 -> 1  (association.py:231)
	} else if (threadIdx.x < 2 * SW_T) {
		const int t = threadIdx.x - SW_T;
		const int64_t gc = (int64_t)bj * SW_T + t;
		const double v = gc < my ? ssy[gc] : 1.0;
		sy[t] = (v == 0.0) ? ncells : v;
	}
	__syncthreads();
	int bad_nf = 0, bad_rng = 0;
	FixAcc facc = nrm_fix_acc();
	const double sqrt_dof = sqrt(fix.dof);
	const FixCol fy = fix_column(fix, gj, my);
	unsigned slow = 0;  // (as in k_assoc_sweep_sym)
#pragma unroll
	for (int i = 0; i < SW_T / 4; i++) {
		const int r = ty + 4 * i;
		const int64_t gi = (int64_t)bi * SW_T + r;
		const bool pair = gi < mx && gj < my;
		double d = dv[i];
		if (fix.fx) d += nrm_fix_corr(sfx[r], fy, fix.top, fix.inv_n);
		const double vx = sx[r], vy = sy[tx];
		const double r2 = (d * d) / (vx * vy);
		bool ok;
		double p = nrm_pvalue_fast(r2, pl, ok);
		double st = d / ncells;
		if (pair) {
			if (!isfinite(r2) || !isfinite(vx) || !isfinite(vy)) bad_nf = 1;
			if (r2 > 1.0 + 1e-8) bad_rng = 1;
			if (!ok) slow |= 1u << i;
			if (fix.fx && fix.budget > 0.0) nrm_fix_note(facc, sfx[r][5], sfx[r][6], r2);
			p_out[(r0 + gi) * ldo + c0 + gj] = (OutT)p;
			stat_out[(r0 + gi) * ldo + c0 + gj] = (OutT)st;
		} else {
			p = 0.0;
			st = 0.0;
		}
		tp[r][tx] = (OutT)p;
		ts[r][tx] = (OutT)st;
	}
	if (slow) {
#pragma unroll 1
		for (int i = 0; i < SW_T / 4; i++) {
			if (!(slow >> i & 1)) continue;
			const int r = ty + 4 * i;
			const int64_t gi = (int64_t)bi * SW_T + r;
			double d = dot[gi * ldd + gj];
			if (fix.fx) d += nrm_fix_corr(sfx[r], fy, fix.top, fix.inv_n);
			const OutT p = (OutT)nrm_pvalue((d * d) / (sx[r] * sy[tx]), pl);
			tp[r][tx] = p;
			p_out[(r0 + gi) * ldo + c0 + gj] = p;
		}
	}
	if (fix.fx && fix.budget > 0.0 && nrm_fix_screen(fix, facc, fy, sqrt_dof)) {  // rare: the exact test, pair by pair
#pragma unroll
		for (int i = 0; i < SW_T / 4; i++) {
			const int r = ty + 4 * i;
			const int64_t gi = (int64_t)bi * SW_T + r;
			if (gi >= mx || gj >= my) continue;
			const double d = dv[i] + nrm_fix_corr(sfx[r], fy, fix.top, fix.inv_n);
			nrm_fix_guard(fix, sfx[r][5], sfx[r][6], fy, (d * d) / (sx[r] * sy[tx]), sqrt_dof, tp[r][tx] != (OutT)0, facc);
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tp / ts are complete; the stores above stay in flight
	const int64_t oj = (int64_t)bi * SW_T + tx;  // mirrored: rows c0 + (block bj), columns r0 + (block bi)
#pragma unroll
	for (int i = 0; i < SW_T / 4; i++) {
		const int r = ty + 4 * i;
		const int64_t oi = (int64_t)bj * SW_T + r;
		if (oi < my && oj < mx) {
			p_out[(c0 + oi) * ldo + r0 + oj] = tp[tx][r];
			stat_out[(c0 + oi) * ldo + r0 + oj] = ts[tx][r];
		}
	}
	if (flags) {
		if (bad_nf) atomicAdd(&flags[0], 1);
		if (bad_rng) atomicAdd(&flags[1], 1);
	}
	fix_finish(fix, flags, facc, pl);
}

// Sweep for the streaming de path (nrm_gram_skinny): one thread per gene.  G[y] = [y C^T (nc) | y X~^T (nx) | 0...],
// ssraw[y] = |y|^2.  |y~|^2 = |y|^2 - a dci a^T with a = y C^T (association.py:226-230 expanded), y~.x~ = y.x~.
#define DS_NZ 32
template <typename OutT>
__global__ void __launch_bounds__(256) k_de_small_sweep(const double* __restrict__ G, const double* __restrict__ ssraw,
														 const double* __restrict__ dci, int nc, int rank_pos,
														 const double* __restrict__ ssx, int nx, int64_t ny, double ncells, double dof,
														 int stat_kind, PvalPlan pl, OutT* __restrict__ p_out, OutT* __restrict__ stat_out,
														 OutT* __restrict__ r_out, OutT* __restrict__ t_out, int64_t ldo,
														 double* __restrict__ ssy_out, double* __restrict__ by_out,
														 int32_t* __restrict__ flags, int const_last) {
	const int64_t y = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (y >= ny) return;
	double g[DS_NZ];
	// columns of G: covariates, then design rows.  const_last: the last covariate (a constant row summed on the vector ALU by
	// nrm_gram_skinny) sits in column 31 and the design rows start one column earlier: put everything back in order
	{
		double raw[DS_NZ];
#pragma unroll
		for (int c = 0; c < DS_NZ; c++) raw[c] = G[y * DS_NZ + c];
#pragma unroll
		for (int c = 0; c < DS_NZ; c++) {
			double v = raw[c];
			if (const_last) {
				if (c == nc - 1)
					v = raw[DS_NZ - 1];
				else if (c >= nc)
					v = c >= 1 ? raw[c - 1] : 0.0;
			}
			g[c] = v;
		}
	}
	double q = 0.0;
	if (rank_pos) {
#pragma unroll
		for (int c = 0; c < DS_NZ; c++) {
			if (c < nc) {
				double b = 0.0;
#pragma unroll
				for (int e = 0; e < DS_NZ; e++)
					if (e < nc) b = fma(dci[c * nc + e], g[e], b);
				q = fma(g[c], b, q);
				if (by_out) by_out[y * nc + c] = b;
			}
		}
	}
	double sy = ssraw[y] - q;
	if (sy < 0.0) sy = 0.0;
	ssy_out[y] = sy;
	const double vy = (sy == 0.0) ? ncells : sy;  // variance 0 -> 1 (association.py:233)
	int bad_nf = 0, bad_rng = 0;
#pragma unroll
	for (int c = 0; c < DS_NZ; c++) {
		const int x = c - nc;
		if (x >= 0 && x < nx) {
			const double d = g[c];
			double vx = ssx[x];
			if (vx == 0.0) vx = ncells;
			const double r2 = (d * d) / (vx * vy);
			if (!isfinite(r2) || !isfinite(vy)) bad_nf = 1;
			if (r2 > 1.0 + 1e-8) bad_rng = 1;
			const int64_t o = (int64_t)x * ldo + y;
			const double pval = nrm_pvalue(r2, pl);
			p_out[o] = (OutT)pval;
			stat_out[o] = (OutT)(stat_kind ? d / vx : d / ncells);
			if (r_out) r_out[o] = (OutT)(d / sqrt(vx * vy));
			if (t_out) {
				const double rc = fmin(r2, 1.0);
				t_out[o] = (OutT)copysign(sqrt(dof * rc / (1.0 - rc)), d);
			}
		}
	}
	if (flags) {
		if (bad_nf) atomicAdd(&flags[0], 1);
		if (bad_rng) atomicAdd(&flags[1], 1);
	}
}

template <typename GT, typename OutT>
__global__ void __launch_bounds__(256) k_alpha(const GT* __restrict__ stat, int64_t ldg, const double* __restrict__ ssx, double ncells,
												const double* __restrict__ bx, const double* __restrict__ by, int64_t nx, int64_t ny,
												int64_t nc, OutT* __restrict__ alpha) {
	int64_t total = nx * ny * nc;
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; i < total; i += stride) {
		int64_t c = i % nc, ij = i / nc;
		int64_t j = ij % ny, x = ij / ny;
		double g = (double)stat[x * ldg + j];
		if (ssx) {  // stat holds the covariance x~.y~/n (return_dot, association.py:1044-1048): gamma = cov * n / |x~|^2
			double vx = ssx[x];
			g *= ncells / (vx == 0.0 ? ncells : vx);
		}
		alpha[i] = (OutT)(by[j * nc + c] - g * bx[x * nc + c]);  // association.py:238-243
	}
}

// ---- C ABI ------------------------------------------------------------------------------------

extern "C" int nrm_pvalues_from_r2(const double* d_r2, int64_t count, double dof, double* d_p, void* stream) {
	nrm_pvalue_plan plan;
	int rc = nrm_pvalue_plan_init(&plan, dof);
	if (rc) return rc;
	if (count <= 0) return NRM_OK;
	NRM_REQUIRE(d_r2 && d_p, "nrm_pvalues_from_r2: null pointer");
	int grid = (int)((count + 255) / 256);
	if (grid > 4096) grid = 4096;
	hipLaunchKernelGGL(k_pvalues_from_r2, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_r2, count, to_dev(plan), d_p);
	return nrm_check_launch("k_pvalues_from_r2");
}

extern "C" int nrm_assoc_sweep_band(const double* d_dot, int64_t ldd, const double* d_ssx, const double* d_ssy, int64_t nx,
									int64_t ny, int64_t n_cells, double dof, int symmetric, int stat_kind, void* d_p,
									void* d_stat, void* d_r, void* d_t, int out_dtype, int64_t ldo, int32_t* d_flags,
									int64_t row0, int64_t row1, int fix_nslices, const double* d_fixx, const double* d_fixy, double guard_tol,
									void* stream) {
	NRM_REQUIRE(nx >= 0 && ny >= 0 && n_cells > 0, "nrm_assoc_sweep: bad sizes");
	NRM_REQUIRE(fix_nslices == 0 || ((fix_nslices == 5 || fix_nslices == 6) && d_fixx && d_fixy && d_flags), "nrm_assoc_sweep: row records need 5 or 6 slices, both tables and flags");
	const FixArgs fix = nrm_fix_args(fix_nslices ? d_fixx : nullptr, d_fixy, fix_nslices, n_cells, dof, guard_tol);
	NRM_REQUIRE(out_dtype == NRM_F32 || out_dtype == NRM_F64, "nrm_assoc_sweep: bad out_dtype");
	NRM_REQUIRE(!symmetric || nx == ny, "nrm_assoc_sweep: symmetric needs nx == ny");
	NRM_REQUIRE(row0 >= 0 && row0 <= row1 && row1 <= nx && row0 % SW_T == 0 && (row1 % SW_T == 0 || row1 == nx),
				"nrm_assoc_sweep_band: rows [row0, row1) must be cut at multiples of %d", SW_T);
	nrm_pvalue_plan plan;
	int rc = nrm_pvalue_plan_init(&plan, dof);
	if (rc) return rc;
	if (nx == 0 || ny == 0 || row0 == row1) return NRM_OK;
	NRM_REQUIRE(d_dot && d_ssx && d_ssy && d_p && d_stat, "nrm_assoc_sweep: null pointer");
	NRM_REQUIRE(ldo >= ny && ldd >= ny, "nrm_assoc_sweep: pitch smaller than row length");
	const int64_t b0 = row0 / SW_T, b1 = (row1 + SW_T - 1) / SW_T;
	if (symmetric && !d_r && !d_t && stat_kind == 0) {
		// upper-triangle blocks of block rows [b0, b1); their mirrored writes land in rows >= row0, columns [row0, row1)
		const int64_t nb = (nx + SW_T - 1) / SW_T;
		const unsigned tiles = (unsigned)((b1 - b0) * nb - (b1 * (b1 - 1) - b0 * (b0 - 1)) / 2);
		// pieces of 32 rows: 0.185 ms on BASELINE configs[1] against 0.196 with whole tiles (tools/k3_time.py), 16 rows gain nothing more
		if (out_dtype == NRM_F64)
			hipLaunchKernelGGL((k_assoc_sweep_sym<double, 32>), dim3(tiles * 2), dim3(256), 0, (hipStream_t)stream, d_dot, ldd, d_ssx, nx, (int)nb,
							   (double)n_cells, to_dev(plan), (double*)d_p, (double*)d_stat, ldo, d_flags, (int)b0, fix);
		else
			hipLaunchKernelGGL((k_assoc_sweep_sym<float, 32>), dim3(tiles * 2), dim3(256), 0, (hipStream_t)stream, d_dot, ldd, d_ssx, nx, (int)nb,
							   (double)n_cells, to_dev(plan), (float*)d_p, (float*)d_stat, ldo, d_flags, (int)b0, fix);
		return nrm_check_launch("k_assoc_sweep_sym");
	}
	dim3 grid((unsigned)((ny + SW_T - 1) / SW_T), (unsigned)(b1 - b0));
	if (out_dtype == NRM_F64)
		hipLaunchKernelGGL(k_assoc_sweep<double>, grid, dim3(256), 0, (hipStream_t)stream, d_dot, ldd, d_ssx, d_ssy, nx, ny,
						   (double)n_cells, dof, symmetric, stat_kind, to_dev(plan), d_p, d_stat, d_r, d_t, ldo, d_flags, (int)b0, fix);
	else
		hipLaunchKernelGGL(k_assoc_sweep<float>, grid, dim3(256), 0, (hipStream_t)stream, d_dot, ldd, d_ssx, d_ssy, nx, ny,
						   (double)n_cells, dof, symmetric, stat_kind, to_dev(plan), d_p, d_stat, d_r, d_t, ldo, d_flags, (int)b0, fix);
	return nrm_check_launch("k_assoc_sweep");
}

extern "C" int nrm_assoc_sweep(const double* d_dot, int64_t ldd, const double* d_ssx, const double* d_ssy, int64_t nx,
							   int64_t ny, int64_t n_cells, double dof, int symmetric, int stat_kind, void* d_p,
							   void* d_stat, void* d_r, void* d_t, int out_dtype, int64_t ldo, int32_t* d_flags,
							   int fix_nslices, const double* d_fixx, const double* d_fixy, double guard_tol, void* stream) {
	return nrm_assoc_sweep_band(d_dot, ldd, d_ssx, d_ssy, nx, ny, n_cells, dof, symmetric, stat_kind, d_p, d_stat, d_r, d_t, out_dtype,
								ldo, d_flags, 0, nx, fix_nslices, d_fixx, d_fixy, guard_tol, stream);
}

extern "C" int nrm_assoc_sweep_mirror(const double* d_dot, int64_t ldd, const double* d_ssx, const double* d_ssy, int64_t mx, int64_t my,
									  int64_t n_cells, double dof, void* d_p, void* d_stat, int out_dtype, int64_t ldo, int64_t r0, int64_t c0,
									  int32_t* d_flags, int fix_nslices, const double* d_fixx, const double* d_fixy, double guard_tol, void* stream) {
	NRM_REQUIRE(fix_nslices == 0 || ((fix_nslices == 5 || fix_nslices == 6) && d_fixx && d_fixy && d_flags), "nrm_assoc_sweep: row records need 5 or 6 slices, both tables and flags");
	const FixArgs fix = nrm_fix_args(fix_nslices ? d_fixx : nullptr, d_fixy, fix_nslices, n_cells, dof, guard_tol);
	NRM_REQUIRE(mx >= 0 && my >= 0 && n_cells > 0 && r0 >= 0 && c0 >= 0 && c0 + my <= r0, "nrm_assoc_sweep_mirror: the rectangle must lie below the diagonal");
	NRM_REQUIRE(out_dtype == NRM_F32 || out_dtype == NRM_F64, "nrm_assoc_sweep: bad out_dtype");
	nrm_pvalue_plan plan;
	int rc = nrm_pvalue_plan_init(&plan, dof);
	if (rc) return rc;
	if (mx == 0 || my == 0) return NRM_OK;
	NRM_REQUIRE(d_dot && d_ssx && d_ssy && d_p && d_stat && ldd >= my && ldo >= r0 + mx, "nrm_assoc_sweep_mirror: null pointer or small pitch");
	dim3 grid((unsigned)((my + SW_T - 1) / SW_T), (unsigned)((mx + SW_T - 1) / SW_T));
	if (out_dtype == NRM_F64)
		hipLaunchKernelGGL(k_assoc_sweep_mirror<double>, grid, dim3(256), 0, (hipStream_t)stream, d_dot, ldd, d_ssx, d_ssy, mx, my, (double)n_cells,
						   to_dev(plan), (double*)d_p, (double*)d_stat, ldo, r0, c0, d_flags, fix);
	else
		hipLaunchKernelGGL(k_assoc_sweep_mirror<float>, grid, dim3(256), 0, (hipStream_t)stream, d_dot, ldd, d_ssx, d_ssy, mx, my, (double)n_cells,
						   to_dev(plan), (float*)d_p, (float*)d_stat, ldo, r0, c0, d_flags, fix);
	return nrm_check_launch("k_assoc_sweep_mirror");
}

extern "C" int nrm_alpha(const void* d_stat, int stat_dtype, int64_t ldg, int stat_kind, const double* d_ssx, int64_t n_cells,
						 const double* d_bx, const double* d_by, int64_t nx, int64_t ny, int64_t nc, void* d_alpha, int out_dtype,
						 void* stream) {
	NRM_REQUIRE(stat_dtype == out_dtype, "nrm_alpha: stat and alpha dtypes must match");
	NRM_REQUIRE(stat_kind == 1 || (stat_kind == 0 && d_ssx && n_cells > 0), "nrm_alpha: stat_kind 0 (covariance) needs d_ssx and n_cells");
	if (nx * ny * nc == 0) return NRM_OK;
	NRM_REQUIRE(d_stat && d_bx && d_by && d_alpha, "nrm_alpha: null pointer");
	const double* ssx = stat_kind == 0 ? d_ssx : nullptr;
	int64_t total = nx * ny * nc;
	int grid = (int)((total + 255) / 256);
	if (grid > 8192) grid = 8192;
	if (out_dtype == NRM_F64)
		hipLaunchKernelGGL((k_alpha<double, double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double*)d_stat,
						   ldg, ssx, (double)n_cells, d_bx, d_by, nx, ny, nc, (double*)d_alpha);
	else
		hipLaunchKernelGGL((k_alpha<float, float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)d_stat, ldg,
						   ssx, (double)n_cells, d_bx, d_by, nx, ny, nc, (float*)d_alpha);
	return nrm_check_launch("k_alpha");
}

extern "C" int nrm_de_small_sweep(const double* d_g, const double* d_ssraw, const double* d_dci, int64_t nc, int rank, const double* d_ssx,
								  int64_t nx, int64_t ny, int64_t n_cells, double dof, int stat_kind, void* d_p, void* d_stat, void* d_r,
								  void* d_t, int out_dtype, int64_t ldo, double* d_ssy, double* d_by, int32_t* d_flags, int const_last, void* stream) {
	NRM_REQUIRE(nx > 0 && ny > 0 && nc >= 0 && nx + nc <= DS_NZ, "nrm_de_small_sweep: needs nx + nc <= %d", DS_NZ);
	NRM_REQUIRE(!const_last || nc >= 1, "nrm_de_small_sweep: const_last needs a covariate");
	NRM_REQUIRE(out_dtype == NRM_F32 || out_dtype == NRM_F64, "nrm_de_small_sweep: bad out_dtype");
	NRM_REQUIRE(d_g && d_ssraw && d_ssx && d_p && d_stat && d_ssy && ldo >= ny, "nrm_de_small_sweep: null pointer or small pitch");
	NRM_REQUIRE(!(rank > 0 && nc > 0) || d_dci, "Unmatching dci dimensions.");
	nrm_pvalue_plan plan;
	int rc = nrm_pvalue_plan_init(&plan, dof);
	if (rc) return rc;
	const int rank_pos = (rank > 0 && nc > 0) ? 1 : 0;
	dim3 grid((unsigned)((ny + 255) / 256));
	if (out_dtype == NRM_F64)
		hipLaunchKernelGGL(k_de_small_sweep<double>, grid, dim3(256), 0, (hipStream_t)stream, d_g, d_ssraw, d_dci, (int)nc, rank_pos, d_ssx,
						   (int)nx, ny, (double)n_cells, dof, stat_kind, to_dev(plan), (double*)d_p, (double*)d_stat, (double*)d_r,
						   (double*)d_t, ldo, d_ssy, d_by, d_flags, const_last);
	else
		hipLaunchKernelGGL(k_de_small_sweep<float>, grid, dim3(256), 0, (hipStream_t)stream, d_g, d_ssraw, d_dci, (int)nc, rank_pos, d_ssx,
						   (int)nx, ny, (double)n_cells, dof, stat_kind, to_dev(plan), (float*)d_p, (float*)d_stat, (float*)d_r,
						   (float*)d_t, ldo, d_ssy, d_by, d_flags, const_last);
	return nrm_check_launch("k_de_small_sweep");
}
