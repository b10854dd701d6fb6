// K1 with the rows RESIDENT ON CHIP between its two phases: every input row is read from HBM exactly once.
//
// k_residualize_v4 (nrm_residualize.hip) sweeps a row twice -- a = x C^T first, then residual -> digits -- and from ~50 000 cells up
// the second sweep misses every cache (256 workgroups x 4 rows x 200 KB .. 4 MB >> 4 MB L2 / 256 MB MALL): 1.6 - 1.8x the
// algorithmic traffic, 0.25 - 0.33 of the HBM roofline (round-3 counters).  Here a work item is 4 rows x ONE SEGMENT of at most
// 24 KB per row, which a workgroup of 256 threads keeps in its registers (96 VGPRs per thread) from the first phase to the
// second.  A row longer than a segment is shared by the nseg workgroups that hold its segments -- a cluster: each posts its partial
// products x C^T (plus max|x|, |x|^2) to a slab in HBM, counts in on the cluster's counter, waits for the others, and adds up the
// nseg partials IN SEGMENT ORDER (every member gets the same bits; no atomics on a result path).  The row records (sums of squares,
// digit statistics: nrm_fix.h) travel the same way; the member that counts in last adds them up and writes them.
//
// No deadlock, whatever the dispatch order or residency: items are handed out by a ticket counter, tickets of a cluster are
// consecutive, and a workgroup posts its partials BEFORE it waits and draws its next ticket only AFTER its cluster has met -- so
// the holder of the smallest ticket that has not posted yet is never waiting for anything.  A wait that lasts 4 s traps (a hung
// box is worse than a failed call).  The counters clean up after themselves (the last member resets its cluster's, the last
// workgroup the ticket counter): the workspace is zeroed once, when it is allocated.
//
// Reference: association.py:224-233 (ccx = dci @ (dc @ dx.T); dx1 = dx - ccx @ dc; mean of squares).
#include "nrm_k1.h"
#include <cstdlib>
#include <algorithm>

#define RR_R 4         // rows per work item
#define RR_NC_MAX 48   // covariates (partials of 4 (nc + 14) doubles per item); beyond: k_residualize_v4
#define RR_STAGE 3072   // doubles of LDS through which the members' partials are gathered (>= 4 (RR_NC_MAX + 2))
#define RR_SEG_MAX 256 // segments per row (cluster size; far below the 512 workgroup slots of the chip)
#define RR_CB 2        // covariates per pass over the registers
#define RR_SLOT 48     // doubles per (wave, row-of-16-lanes) slot of the workgroup reductions (>= 4 rows x 11 record entries)

template <typename T>
struct RRGeom {
	static constexpr int G = sizeof(T) == 4 ? 6 : 3;  // groups of 4 cells per thread and row: 24 KB of a row per item (96 VGPRs)
};

template <typename T>
__device__ __forceinline__ void rr_ld4(const T* p, T (&v)[4]);
template <>
__device__ __forceinline__ void rr_ld4<float>(const float* p, float (&v)[4]) {
	const float4 t = *reinterpret_cast<const float4*>(p);
	v[0] = t.x;
	v[1] = t.y;
	v[2] = t.z;
	v[3] = t.w;
}
template <>
__device__ __forceinline__ void rr_ld4<double>(const double* p, double (&v)[4]) {
	const double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2);
	v[0] = a.x;
	v[1] = a.y;
	v[2] = b.x;
	v[3] = b.y;
}

// A register-resident input value as a double, through an empty asm: the conversion of an fp32 value is then redone where it is
// used instead of being hoisted out of the covariate loops (128 values as doubles are the whole register file).
template <typename T>
__device__ __forceinline__ double rr_val(T v) {
	asm volatile("" : "+v"(v));
	return (double)v;
}

// Sum / maximum of a double over the 64 lanes without LDS round trips: four DPP steps inside every row of 16 lanes (two 32-bit moves
// and one fp64 operation each), then the four row results through scalar registers.  Every lane gets the result.
template <int CTRL>
__device__ __forceinline__ double rr_dpp(double v) {
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rr_row_sum(double v) {  // over the 16 lanes of a DPP row; every lane of the row gets it
	v += rr_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
	v += rr_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
	v += rr_dpp<0x141>(v);  // row_half_mirror
	v += rr_dpp<0x140>(v);  // row_mirror
	return v;
}
__device__ __forceinline__ double rr_row_max(double v) {
	v = fmax(v, rr_dpp<0xB1>(v));
	v = fmax(v, rr_dpp<0x4E>(v));
	v = fmax(v, rr_dpp<0x141>(v));
	v = fmax(v, rr_dpp<0x140>(v));
	return v;
}
// fold the 16 (wave, row) partials of a workgroup in a fixed order
__device__ __forceinline__ double rr_sum16(const double (*p)[RR_SLOT], int o) {
	double v[4];
#pragma unroll
	for (int w = 0; w < 4; w++) v[w] = (p[4 * w][o] + p[4 * w + 1][o]) + (p[4 * w + 2][o] + p[4 * w + 3][o]);
	return (v[0] + v[1]) + (v[2] + v[3]);
}
__device__ __forceinline__ double rr_max16(const double (*p)[RR_SLOT], int o) {
	double v = p[0][o];
#pragma unroll
	for (int w = 1; w < 16; w++) v = fmax(v, p[w][o]);
	return v;
}

// slab traffic between the members of a cluster: agent-scope (sc1) stores and loads, which no L1 and no other XCD's L2 keeps
__device__ __forceinline__ void rr_post(double* p, double v) {
	__hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double rr_fetch(const double* p) {
	return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(const_cast<double*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// thread 0 of a workgroup, after the workgroup's posts have drained (s_waitcnt vmcnt(0) in every wave + barrier): count in, wait
// for the cluster, acquire
__device__ __forceinline__ void rr_meet(int* ctr, int members) {
	__hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const long long t0 = wall_clock64();
	while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < members) {
		__builtin_amdgcn_s_sleep(2);
		if (wall_clock64() - t0 > 400000000ll) __builtin_trap();  // 4 s of the 100 MHz clock
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T, int NS>
__global__ void __launch_bounds__(256, 2) k_residualize_res(const T* __restrict__ x, int64_t rows, int64_t n, int64_t ldx, const double* __restrict__ c, int nc,
															 int64_t ldc, const double* __restrict__ dci, int active, double* __restrict__ ss,
															 double* __restrict__ coef, QuantOut qo, int* __restrict__ ctr, double* __restrict__ part, int stride,
															 int nseg, int gseg, int ngroups, long long* __restrict__ dbg, int ablate, int teams) {
	constexpr int R = RR_R, G = RRGeom<T>::G, CB = RR_CB, NP = NS - 1, NREC = 2 * NP + 1, B = 8 * NS - 2;
	extern __shared__ double s_dyn[];
	__shared__ double s_w[16][RR_SLOT];  // one slot per (wave, row of 16 lanes)
	__shared__ double s_fin[R * NREC];
	__shared__ double s_stage[RR_STAGE];
	__shared__ double s_xm[R], s_xq[R], s_mx[R];
	__shared__ int s_sh[R], s_loose, s_item, s_last;
	const int tid = threadIdx.x, lane = tid & 63, slot = tid >> 4;
	const bool first = (lane & 15) == 0;  // the lane that writes its row's partial
	const int64_t items = (int64_t)ngroups * nseg;
	// pseudo-inverse and covariate maxima: once per workgroup into LDS behind the OLS tables
	double* const s_dci = s_dyn + (size_t)2 * R * ((nc + CB - 1) / CB * CB > 0 ? (nc + CB - 1) / CB * CB : CB);
	double* const s_cmax = s_dci + (size_t)nc * nc;
	if (active) {
		for (int i = tid; i < nc * nc; i += 256) s_dci[i] = dci[i];
		for (int i = tid; i < nc; i += 256) s_cmax[i] = qo.cmax ? qo.cmax[i] : 0.0;
	}

	// Items come from the ticket counter -- or, teams > 0 (NRM_K1_TEAMS=1, an experiment: it needs all teams * nseg workgroups of the grid
	// resident at once, which a plain launch cannot promise), from a fixed plan: workgroup b is member b % nseg of team b / nseg and
	// takes that segment of row groups team, team + teams, ...  The members of a cluster then run in lock step instead of arriving
	// as their previous items happen to end.
	const int64_t stride_items = (int64_t)teams * nseg;
	if (tid == 0) s_item = teams ? (int)blockIdx.x : atomicAdd(ctr, 1);
	__syncthreads();
	int64_t item = s_item;
	while (item < items) {
		// (the arguments pass through an empty asm at the top of every item: what is derived from them is recomputed per item on the
		// scalar unit instead of being hoisted out of the loop and kept -- spilled -- in vector registers across it)
		asm volatile("" : "+s"(n), "+s"(nc), "+s"(ldc), "+s"(ldx), "+s"(rows), "+s"(nseg), "+s"(gseg), "+s"(active), "+s"(stride));
		asm volatile("" : "+s"(qo.nks), "+s"(qo.cks), "+s"(qo.chunk_bytes), "+s"(qo.plane_bytes));  // (not the pointers: they would turn into flat ones)
		const int64_t kq = qo.nks * 32;
		const int64_t klast = n - 4;  // (n % 4 == 0: the launcher sends other rows to k_residualize_v4)
		const int gtotal = (int)((kq + 1023) / 1024);
		const int na = nc + 2;  // a slab row: nc products, max|x|, |x|^2
		const int ncp = (nc + CB - 1) / CB * CB;  // covariates padded to whole passes (b of a padding covariate is 0)
		double* const ta = s_dyn;                                  // [R][ncp]  x_i C^T (whole rows)
		double* const tb = s_dyn + (size_t)R * (ncp > 0 ? ncp : CB);  // [R][ncp]  (x_i C^T) dci
		const bool bounded = active && qo.cmax != nullptr;
		const int group = (int)(item / nseg), seg = (int)(item - (int64_t)group * nseg);
		const int64_t row0 = (int64_t)group * R, k0 = (int64_t)seg * gseg * 1024 + tid * 4;
		const int gcount = gtotal - seg * gseg < gseg ? gtotal - seg * gseg : gseg;  // groups of 1024 cells in this segment
		// profiling aid (nrm_k1_debug_buffer): 8 time stamps of the 100 MHz clock per item
		auto stamp = [&](int i) {
			if (dbg && tid == 0) dbg[item * 8 + i] = wall_clock64();
		};
		stamp(0);
		int* const gc = ctr + 4 + 4 * group;  // the cluster's counters: first meeting, meeting of the true maxima, row records
		double* const mine = part + item * stride;
		const double* const slab = part + (int64_t)group * nseg * stride;
		// ---- the item's cells -> registers (raw input type), all loads in flight at once; zeros past the row and for padding rows ----
		T d[R][G][4];
#pragma unroll
		for (int g = 0; g < G; g++) {
			const int64_t k = k0 + (int64_t)g * 1024;
#pragma unroll
			for (int r = 0; r < R; r++) {
				const bool live = row0 + r < rows;
				const T* xr = x + (live ? (row0 + r) : 0) * ldx;
				if (g < gcount && live && k < n)
					rr_ld4<T>(xr + k, d[r][g]);
				else
					d[r][g][0] = d[r][g][1] = d[r][g][2] = d[r][g][3] = (T)0;
			}
		}
		// covariates c0 .. c0 + CB - 1 (clamped to the last one: a padding covariate's b is 0, its product unused) at the 4 cells of
		// group g (clamped to the last cells of the row: the data there are zeros)
		auto cov = [&](int c0, int g, double (&cv)[CB][4]) {
			int64_t k = k0 + (int64_t)g * 1024;
			k = k < klast ? k : klast;
			if (ablate & 2) k = tid * 4;  // (ablation: every covariate load from the same 8 KB -- L1 / L2 hits)
#pragma unroll
			for (int q = 0; q < CB; q++) {
				const int qq = c0 + q < nc ? c0 + q : nc - 1;
				Vec4Load<double>::ld(c + (int64_t)qq * ldc + k, cv[q]);
			}
		};
		// fold `count` (<= 256) consecutive slab values of every member, in segment order: thread i < count gets element i.  All threads
		// fetch (independent loads, staged through LDS); is_max(i) says whether element i is a maximum or a sum.
		auto gather = [&](int off, int count, auto is_max, auto put) {
			const int sb = RR_STAGE / count;
			double acc = 0.0;
			for (int s0 = 0; s0 < nseg; s0 += sb) {
				const int ns = nseg - s0 < sb ? nseg - s0 : sb;
#pragma unroll 4
				for (int j = tid; j < ns * count; j += 256) {
					const int s = j / count, i = j - s * count;
					s_stage[j] = rr_fetch(slab + (int64_t)(s0 + s) * stride + off + i);
				}
				__syncthreads();
				if (tid < count)
					for (int s = 0; s < ns; s++) {
						const double t = s_stage[s * count + tid];
						acc = is_max(tid) ? fmax(acc, t) : acc + t;
					}
				__syncthreads();
			}
			if (tid < count) put(tid, acc);
		};
		// ---- phase A: this segment's share of a = x C^T, max|x|, |x|^2 ----
		{
			double xmax[R], xsq[R];
#pragma unroll
			for (int r = 0; r < R; r++) {
				xmax[r] = xsq[r] = 0.0;
#pragma unroll
				for (int g = 0; g < G; g++)
#pragma unroll
					for (int i = 0; i < 4; i++) {
						const double v = (double)d[r][g][i];
						xmax[r] = fmax(xmax[r], fabs(v));
						xsq[r] = fma(v, v, xsq[r]);
					}
			}
#pragma unroll
			for (int r = 0; r < R; r++) {
				const double m = rr_row_max(xmax[r]);
				const double q = rr_row_sum(xsq[r]);
				if (first) {
					s_w[slot][r * (CB + 2) + CB] = m;
					s_w[slot][r * (CB + 2) + CB + 1] = q;
				}
			}
			__syncthreads();
			stamp(1);
			if (tid < R) {
				const int o = tid * (CB + 2) + CB;
				const double m = rr_max16(s_w, o);
				const double q = rr_sum16(s_w, o + 1);
				if (nseg == 1) {
					s_xm[tid] = m;
					s_xq[tid] = q;
				} else {
					rr_post(mine + tid * na + nc, m);
					rr_post(mine + tid * na + nc + 1, q);
				}
			}
			if (active) {
				// CB covariates per pass over the registers; the covariate values of the next group (or of the next pass) are fetched
				// while this group's products are taken
				double cvn[CB][4];
				cov(0, 0, cvn);
				for (int c0 = 0; c0 < nc; c0 += CB) {
					double acc[R][CB];
#pragma unroll
					for (int r = 0; r < R; r++)
#pragma unroll
						for (int q = 0; q < CB; q++) acc[r][q] = 0.0;
#pragma unroll
					for (int g = 0; g < G; g++) {
						if (g < gcount) {
							double cv[CB][4];
#pragma unroll
							for (int q = 0; q < CB; q++)
#pragma unroll
								for (int i = 0; i < 4; i++) cv[q][i] = cvn[q][i];
							if (g + 1 < gcount)
								cov(c0, g + 1, cvn);
							else if (c0 + CB < nc)
								cov(c0 + CB, 0, cvn);
#pragma unroll
							for (int r = 0; r < R; r++)
#pragma unroll
								for (int i = 0; i < 4; i++) {
									const double xv = rr_val(d[r][g][i]);
#pragma unroll
									for (int q = 0; q < CB; q++) acc[r][q] = fma(xv, cv[q][i], acc[r][q]);
								}
						}
					}
					__syncthreads();  // (s_w of the pass before has been read)
#pragma unroll
					for (int r = 0; r < R; r++)
#pragma unroll
						for (int q = 0; q < CB; q++) {
							const double v = rr_row_sum(acc[r][q]);
							if (first) s_w[slot][r * (CB + 2) + q] = v;
						}
					__syncthreads();
					if (tid < R * CB) {
						const int r = tid / CB, q = tid % CB;
						if (c0 + q < nc) {
							const int o = r * (CB + 2) + q;
							const double v = rr_sum16(s_w, o);
							if (nseg == 1)
								ta[r * ncp + c0 + q] = v;
							else
								rr_post(mine + r * na + c0 + q, v);
						}
					}
				}
			}
		}
		// ---- the cluster meets: every member adds up the nseg partials in segment order ----
		stamp(2);
		if (nseg > 1) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			if (tid == 0) rr_meet(gc, nseg);
			stamp(3);
			__syncthreads();
			gather(0, R * na, [&](int i) { return i % na == nc; },
				   [&](int i, double v) {
					   const int r = i / na, q = i - r * na;
					   if (q < nc) {
						   if (active) ta[r * ncp + q] = v;
					   } else if (q == nc)
						   s_xm[r] = v;
					   else
						   s_xq[r] = v;
				   });
		}
		__syncthreads();
		stamp(4);
		if (tid == 0) s_item = teams ? (int)(item + stride_items) : atomicAdd(ctr, 1);  // the next ticket, drawn only now: it cannot lie in this cluster (see the header)
		if (active) {
			for (int i = tid; i < R * ncp; i += 256) {
				const int r = i / ncp, q = i - r * ncp;
				double v = 0.0;
				if (q < nc) {
					for (int e = 0; e < nc; e++) v = fma(s_dci[q * nc + e], ta[r * ncp + e], v);
					if (coef && seg == 0 && row0 + r < rows) coef[(row0 + r) * nc + q] = v;
				}
				tb[i] = v;
			}
		}
		if (tid == 0) s_loose = 0;
		__syncthreads();
		// ---- fixed-point scale of each row: 2^e >= the largest |residual| (the rule of k_residualize_v4) ----
		if (tid < R) {
			double m = s_xm[tid];
			if (bounded) {
				const double sq = s_xq[tid];
				double proj = 0.0;
				for (int q = 0; q < nc; q++) {
					m = fma(fabs(tb[tid * ncp + q]), s_cmax[q], m);
					proj = fma(ta[tid * ncp + q], tb[tid * ncp + q], proj);
				}
				const double est = sq - proj;  // |x~|^2 up to cancellation: trusted only while it is a fair share of |x|^2
				if (row0 + tid < rows && !(est > 1e-8 * sq && m * m * (double)n <= (RES_LOOSE * RES_LOOSE) * est)) s_loose = 1;
			} else if (active)
				s_loose = 1;  // no maxima of the covariates from the caller: look
			s_mx[tid] = m;
		}
		__syncthreads();
		// residual of group g from the registers; cvn holds the covariate values of its first pass on entry and those of the next
		// group's first pass on exit
		auto residual = [&](int g, double (&cvn)[CB][4], double (&v)[R][4]) {
#pragma unroll
			for (int r = 0; r < R; r++)
#pragma unroll
				for (int i = 0; i < 4; i++) v[r][i] = rr_val(d[r][g][i]);
			if (active) {
				for (int c0 = 0; c0 < nc; c0 += CB) {
					double cv[CB][4];
#pragma unroll
					for (int q = 0; q < CB; q++)
#pragma unroll
						for (int i = 0; i < 4; i++) cv[q][i] = cvn[q][i];
					if (c0 + CB < nc)
						cov(c0 + CB, g, cvn);
					else if (g + 1 < gcount)
						cov(0, g + 1, cvn);
#pragma unroll
					for (int r = 0; r < R; r++)
#pragma unroll
						for (int q = 0; q < CB; q++) {
							const double t = tb[r * ncp + c0 + q];
#pragma unroll
							for (int i = 0; i < 4; i++) v[r][i] = fma(-t, cv[q][i], v[r][i]);
						}
				}
			}
		};
		if (s_loose) {  // (one decision for the cluster: every member sees the same sums)
			double rmax[R];
#pragma unroll
			for (int r = 0; r < R; r++) rmax[r] = 0.0;
			double cvn[CB][4];
			cov(0, 0, cvn);
#pragma unroll
			for (int g = 0; g < G; g++) {
				if (g < gcount) {
					double v[R][4];
					residual(g, cvn, v);
#pragma unroll
					for (int r = 0; r < R; r++)
#pragma unroll
						for (int i = 0; i < 4; i++) rmax[r] = fmax(rmax[r], k0 + (int64_t)g * 1024 < n ? fabs(v[r][i]) : 0.0);
				}
			}
#pragma unroll
			for (int r = 0; r < R; r++) {
				const double m = rr_row_max(rmax[r]);
				if (first) s_w[slot][r] = m;
			}
			__syncthreads();
			if (tid < R) {
				const double m = rr_max16(s_w, tid);
				if (nseg == 1)
					s_mx[tid] = m;
				else
					rr_post(mine + R * na + tid, m);
			}
			if (nseg > 1) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__syncthreads();
				if (tid == 0) rr_meet(gc + 1, nseg);
				__syncthreads();
				gather(R * na, R, [](int) { return true; }, [&](int i, double v) { s_mx[i] = v; });
			}
			__syncthreads();
		}
		if (tid < R) {
			const double m = s_mx[tid];
			int e = 0;
			if (m > 0.0 && m < INFINITY) (void)frexp(m, &e);  // m = f 2^e, f in [0.5, 1): every |residual| < 2^e
			s_sh[tid] = e - B;
			if (seg == 0) qo.exps[row0 + tid] = e - B;
		}
		__syncthreads();
		stamp(5);
		// ---- phase B: residual from the registers -> sum of squares, digits (layout: nrm_gram_i8.hip), digit statistics ----
		double sq[R];
		int dsum[R][NP];
		unsigned dsq[R][NP];
		int sh[R], flip[R];
		char* qrow[R];
#pragma unroll
		for (int r = 0; r < R; r++) {
			sq[r] = 0.0;
#pragma unroll
			for (int s = 0; s < NP; s++) {
				dsum[r][s] = 0;
				dsq[r][s] = 0u;
			}
			sh[r] = s_sh[r];
			const int64_t row = row0 + r;
			const int rr = (int)(row & 31);
			qrow[r] = qo.q + ((row >> 5) * qo.cks) * 1024 + (2 * rr) * 16;
			flip[r] = (rr >> 3) & 1;
		}
		{
			double cvn[CB][4];
			if (active) cov(0, 0, cvn);
#pragma unroll
			for (int g = 0; g < G; g++) {
				if (g < gcount) {
					const int64_t k = k0 + (int64_t)g * 1024;
					double v[R][4];
					residual(g, cvn, v);
					if (k < kq) {
						const int kk = (int)(k & 31);
						const int ks_all = (int)(k >> 5), chunk = ks_all / (int)qo.cks;
						const int64_t ks = (int64_t)(ks_all - chunk * (int)qo.cks) + chunk * (qo.chunk_bytes >> 10);  // in KB images from q
#pragma unroll
						for (int r = 0; r < R; r++) {
#pragma unroll
							for (int i = 0; i < 4; i++) {
								if (k >= n) v[r][i] = 0.0;  // (cells past the row: zero digits)
								sq[r] = fma(v[r][i], v[r][i], sq[r]);
							}
							unsigned w[NS];
							nrm_digits4<NS>(v[r], sh[r], w);
							char* dst = qrow[r] + ks * 1024 + (((kk >> 4) ^ flip[r]) << 4) + (kk & 15);
							if (!(ablate & 1)) {
#pragma unroll
								for (int s = 0; s < NS; s++) *reinterpret_cast<unsigned*>(dst + s * qo.plane_bytes) = w[s];
							}
#pragma unroll
							for (int s = 0; s < NP; s++) {
								dsum[r][s] = __builtin_amdgcn_sdot4((int)w[s], 0x01010101, dsum[r][s], false);
								dsq[r][s] = (unsigned)__builtin_amdgcn_sdot4((int)w[s], (int)w[s], (int)dsq[r][s], false);
							}
						}
					}
				}
			}
		}
		stamp(6);
		// ---- row records: sum of squares and digit statistics of this segment -> the row's ----
#pragma unroll
		for (int r = 0; r < R; r++) {
			const double v = rr_row_sum(sq[r]);
			if (first) s_w[slot][r * NREC] = v;
#pragma unroll
			for (int s = 0; s < NP; s++) {
				const int a = row16_sum(dsum[r][s]);
				const unsigned b = (unsigned)row16_sum((int)dsq[r][s]);
				if (first) {
					s_w[slot][r * NREC + 1 + s] = (double)a;  // (integers far below 2^53: exact)
					s_w[slot][r * NREC + 1 + NP + s] = (double)b;
				}
			}
		}
		__syncthreads();
		if (tid < R * NREC) {
			const double v = rr_sum16(s_w, tid);
			if (nseg == 1)
				s_fin[tid] = v;
			else
				rr_post(mine + R * (na + 1) + tid, v);
		}
		bool finish = nseg == 1;
		if (nseg > 1) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			if (tid == 0) {
				const int before = __hip_atomic_fetch_add(gc + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				s_last = before == nseg - 1;
				if (s_last) {
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				}
			}
			__syncthreads();
			finish = s_last != 0;
			if (finish)  // the member that counted in last adds up the records, in segment order
				gather(R * (na + 1), R * NREC, [](int) { return false; }, [&](int i, double v) { s_fin[i] = v; });
		}
		__syncthreads();
		if (finish && tid < R) {
			const double ssr = s_fin[tid * NREC];
			ss[row0 + tid] = ssr;
			if (qo.fix) {
				double S[5] = {0, 0, 0, 0, 0}, Q[5] = {0, 0, 0, 0, 0};
#pragma unroll
				for (int s = 0; s < NP; s++) {
					S[s] = s_fin[tid * NREC + 1 + s];
					Q[s] = s_fin[tid * NREC + 1 + NP + s];
				}
				nrm_fix_record<NS>(qo.fix + (row0 + tid) * NRM_FIX_STRIDE, S, Q, s_sh[tid], ssr, (double)n);
			}
			if (nseg > 1 && tid == 0) {  // every member is past both meetings: the counters are free for the next launch
				__hip_atomic_store(gc, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(gc + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(gc + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		__syncthreads();
		stamp(7);
		item = s_item;
	}
	// every workgroup draws exactly one ticket past the end; the one that draws the last of them resets the ticket counter
	if (tid == 0 && !teams) {
		const int od = __hip_atomic_fetch_add(ctr + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (od == (int)gridDim.x - 1) {
			__hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(ctr + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

// segments per row and groups of 1024 cells per segment for rows of k_ext cells (digit planes included: k_ext = 32 * k-steps)
static void rr_geometry(int x_dtype, int64_t k_ext, int* nseg, int* gseg) {
	const int gmax = x_dtype == NRM_F32 ? RRGeom<float>::G : RRGeom<double>::G;
	const int64_t groups = (k_ext + 1023) / 1024;
	int64_t s = (groups + gmax - 1) / gmax;
	const int64_t g = (groups + s - 1) / s;
	s = (groups + g - 1) / g;
	*nseg = (int)s;
	*gseg = (int)g;
}

static int64_t rr_k_ext(int64_t n, int64_t chunk_ksteps) {
	int64_t nks = ((n + 15) / 16 * 16 + 31) / 32;
	if (chunk_ksteps > 0) nks = (nks + chunk_ksteps - 1) / chunk_ksteps * chunk_ksteps;
	return nks * 32;
}

// true: the resident kernel takes this shape
bool nrm_k1_res_applies(int x_dtype, int64_t n, int64_t nc, int64_t chunk_ksteps) {
	if (nc > RR_NC_MAX || n % 4 != 0 || n < 4) return false;
	int nseg, gseg;
	rr_geometry(x_dtype, rr_k_ext(n, chunk_ksteps), &nseg, &gseg);
	return nseg <= RR_SEG_MAX;
}

extern "C" int64_t nrm_residualize_workspace_bytes(int x_dtype, int64_t rows_pad, int64_t n, int64_t nc, int64_t chunk_ksteps) {
	if (rows_pad <= 0 || n <= 0 || nc < 0 || !nrm_k1_res_applies(x_dtype, n, nc, chunk_ksteps)) return 0;
	int nseg, gseg;
	rr_geometry(x_dtype, rr_k_ext(n, chunk_ksteps), &nseg, &gseg);
	const int64_t groups = (rows_pad + RR_R - 1) / RR_R;
	const int64_t ints = (4 + 4 * groups + 3) / 4 * 4;  // 16-byte aligned slabs behind the counters
	return ints * 4 + (nseg > 1 ? groups * nseg * (int64_t)RR_R * (nc + 14) * 8 : 0);
}

// Profiling aid: a device buffer of 8 int64 per work item (rows_pad / 4 x segments) that the next launches fill with time stamps of
// the 100 MHz clock (start, rows loaded, products posted, cluster met, partials gathered, scale decided, digits written, done);
// nullptr switches it off.  Not part of the product path.
static long long* g_rr_dbg = nullptr;
static int g_rr_ablate = 0;  // bit 0: no digit stores, bit 1: covariate loads from one 8 KB window (timing experiments only: wrong results)
extern "C" int nrm_k1_debug_buffer(void* d_stamps) {
	g_rr_dbg = (long long*)d_stamps;
	const char* a = getenv("NRM_K1_ABLATE");
	g_rr_ablate = a ? atoi(a) : 0;
	return NRM_OK;
}

static int rr_slots() {  // workgroup slots of the device at two per CU (256 VGPRs per thread)
	static int cached[64] = {0};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
	if (!cached[dev]) {
		int cus = 0;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
		cached[dev] = 2 * cus;
	}
	return cached[dev];
}

int nrm_k1_res_launch(const void* d_x, int x_dtype, int64_t rows, int64_t n, int64_t ldx, const double* d_c, int nc, int64_t ldc, const double* d_dci,
					  int active, int64_t rows_pad, double* d_ss, double* d_coef, int nslices, const QuantOut& qo, void* d_work, int64_t work_bytes,
					  int64_t chunk_ksteps, hipStream_t st) {
	int nseg, gseg;
	rr_geometry(x_dtype, qo.nks * 32, &nseg, &gseg);
	NRM_REQUIRE(nseg <= RR_SEG_MAX && nc <= RR_NC_MAX && rows_pad % RR_R == 0, "nrm_residualize: shape outside the resident kernel");
	NRM_REQUIRE(d_work && ((uintptr_t)d_work % 16 == 0) && work_bytes >= nrm_residualize_workspace_bytes(x_dtype, rows_pad, n, nc, chunk_ksteps),
				"nrm_residualize: workspace smaller than nrm_residualize_workspace_bytes()");
	const int64_t groups = rows_pad / RR_R;
	const int64_t ints = (4 + 4 * groups + 3) / 4 * 4;
	int* ctr = (int*)d_work;
	double* part = (double*)((char*)d_work + ints * 4);
	const int stride = RR_R * (nc + 14);
	const int64_t items = groups * nseg;
	const int slots = rr_slots();
	// NRM_K1_TEAMS=1: whole teams of nseg workgroups, as many as fit the slots (see the kernel)
	static const bool want_teams = getenv("NRM_K1_TEAMS") && atoi(getenv("NRM_K1_TEAMS")) > 0;
	int teams = 0;
	if (want_teams && nseg > 1 && nseg <= slots) teams = (int)std::min<int64_t>(slots / nseg, groups);
	const dim3 grid(teams ? (unsigned)(teams * nseg) : (unsigned)(items < slots ? items : slots));
	const int ncp = (nc + RR_CB - 1) / RR_CB * RR_CB;
	const size_t lds = ((size_t)2 * RR_R * (ncp > 0 ? ncp : RR_CB) + (size_t)nc * nc + nc) * sizeof(double);
#define RR_GO(T, NS)                                                                                                                            \
	hipLaunchKernelGGL((k_residualize_res<T, NS>), grid, dim3(256), lds, st, (const T*)d_x, rows, n, ldx, d_c, nc, ldc, d_dci, active, d_ss, d_coef, qo, \
					   ctr, part, stride, nseg, gseg, (int)groups, g_rr_dbg, g_rr_dbg ? g_rr_ablate : 0, teams)
	if (x_dtype == NRM_F32) {
		if (nslices == 6)
			RR_GO(float, 6);
		else
			RR_GO(float, 5);
	} else {
		if (nslices == 6)
			RR_GO(double, 6);
		else
			RR_GO(double, 5);
	}
#undef RR_GO
	return nrm_check_launch("k_residualize_res");
}
