// Whole-problem host entries beyond dense single=0 (numpy buffers in, numpy buffers out, no torch): what a maintainer of the reference binds at
// the association_tests seam (association.py:761-771,1093) for the CRISPR-screen calls of BASELINE configs[3] --
//   * the sparse-design form of single=0 de inside nrm_association_tests_host (association.py:224-235 for a design with few entries),
//   * nrm_association_tests_single1_host   (`normalisr de -m single`:    association.py:263-390,911-925),
//   * nrm_association_tests_single4_host   (`normalisr de -m covariate`: association.py:421-576,926-980, full-rank designs),
//   * nrm_binnet_host                      (binnet.py:134-173)
// -- the same kernels the Python host (normalisr_amd/de_sparse.py, single1.py, single4.py, binnet.py) drives through the device-pointer
// entries, sequenced here in C++ with the library's own scratch pool.  Round 4 had only the dense path behind the C seam: configs[3] ran in
// 15.9 ms there against 2.6 ms through the package.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "nrm_host_entry.h"
#include "nrm_design.h"

int NrmDesignLists::build(const void* d_x, int x_dtype, int64_t nx, int64_t n, bool want_ell, double max_density, hipStream_t st) {
	nslots = nrm_round_up(nx, 64);
	ngroups = nslots / 64;
	nch = (n + DS_CH - 1) / DS_CH;
	NRM_TRY(cnt.alloc((size_t)nch * nslots * 4));
	NRM_TRY(coff.alloc((size_t)nch * nslots * 4));
	NRM_TRY(info.alloc(8 * sizeof(int64_t)));
	NRM_TRY(row_ptr.alloc((size_t)(nx + 1) * 8));
	NRM_TRY(slot2x.alloc((size_t)nslots * 4));
	if (want_ell) {
		NRM_TRY(sig.alloc((size_t)nch * nslots * 4));
		NRM_TRY(pos.alloc((size_t)nch * nslots * 4));
		NRM_TRY(w.alloc((size_t)nch * ngroups * 4));
		NRM_TRY(base.alloc((size_t)nch * ngroups * 8));
	}
	NRM_TRY(nrm_design_count(d_x, x_dtype, nx, n, n, cnt.as<int32_t>(), nslots, info.as<int64_t>(), st));
	NRM_TRY(nrm_design_plan(cnt.as<int32_t>(), nx, n, nslots, sig.as<int32_t>(), pos.as<int32_t>(), w.as<int32_t>(), base.as<int64_t>(), row_ptr.as<int64_t>(),
							coff.as<int32_t>(), slot2x.as<int32_t>(), info.as<int64_t>(), st));
	int64_t h[8];
	NRM_HIP(hipMemcpyAsync(h, info.p, sizeof(h), hipMemcpyDeviceToHost, st));
	NRM_HIP(hipStreamSynchronize(st));
	nnz = h[0];
	padded = h[1];
	bits = (int)h[2];
	binary = !(bits & NRM_DESIGN_NOTONE);
	ok = nnz > 0 && (double)nnz <= max_density * (double)nx * (double)n;
	if (!ok) return NRM_OK;
	NRM_TRY(cells.alloc((size_t)nnz * 4));
	if (!binary) NRM_TRY(row_vals.alloc((size_t)nnz * 8));
	if (want_ell) {
		NRM_TRY(ell.alloc((size_t)(padded > 8 ? padded : 8) * 2));
		if (!binary) NRM_TRY(ellv.alloc((size_t)(padded > 8 ? padded : 8) * 8));
	}
	return nrm_design_fill(d_x, x_dtype, nx, n, n, nslots, pos.as<int32_t>(), w.as<int32_t>(), base.as<int64_t>(), row_ptr.as<int64_t>(), coff.as<int32_t>(),
						   ell.as<int16_t>(), ellv.as<double>(), cells.as<int32_t>(), row_vals.as<double>(), binary ? 1 : 0, st);
}

namespace {

// a constant covariate row (the intercept): its index and value, or -1
int constant_row(const double* c64, int64_t nc, int64_t n, double* value) {
	for (int64_t c = 0; c < nc; c++) {
		const double v = c64[c * n];
		if (v == 0.0) continue;
		bool all = true;
		for (int64_t k = 1; k < n && all; k++) all = c64[c * n + k] == v;
		if (all) {
			*value = v;
			return (int)c;
		}
	}
	*value = 0.0;
	return -1;
}

// variances = ss / n with the 0 -> 1 rule (association.py:230-233), cast to the output dtype
int emit_var(const double* d_ss, int64_t cnt, int64_t n, void* h_out, int out_dtype) {
	std::vector<double> hs((size_t)cnt);
	NRM_HIP(hipMemcpy(hs.data(), d_ss, (size_t)cnt * 8, hipMemcpyDeviceToHost));
	for (int64_t i = 0; i < cnt; i++) {
		double v = hs[(size_t)i] / (double)n;
		if (v == 0.0) v = 1.0;
		if (out_dtype == NRM_F64)
			((double*)h_out)[i] = v;
		else
			((float*)h_out)[i] = (float)v;
	}
	return NRM_OK;
}

// x~ . y~ for every (design row, expression row) through the sparse-design kernels: d_dot (nxp, nyp) or, by_gene, (nyp, nxp); the rows' sums with
// the covariates inside the kernel for up to nrm_de_sparse_fused_covariates() of them, by the stream kernel of single=1 otherwise
int sparse_products(NrmDesignLists& L, const void* d_y, int y_dtype, int64_t ny, int64_t n, const double* d_c, int64_t ncu, int ci, double cval, const double* d_dci,
					const double* d_bx, int64_t ldb, double* d_dot, int64_t ldd, int by_gene, double* d_ssy, double* d_coefy, int32_t* d_flags, hipStream_t st) {
	DevBuf common, ct, code;
	NRM_TRY(common.alloc((size_t)(ncu + 1) * ny * 8));
	const bool fused = ncu - (ci >= 0 ? 1 : 0) <= nrm_de_sparse_fused_covariates();
	if (fused) {
		NRM_TRY(ct.alloc((size_t)nrm_de_sparse_ct_doubles(n, ncu, ci) * 8));
	} else {
		NRM_TRY(code.alloc((size_t)n * 4));
		NRM_TRY(nrm_fill_i32(code.p, NRM_S1_COMMON, n, st));
		NRM_TRY(nrm_single1_stream(d_y, y_dtype, n, d_c, n, ncu, code.as<int32_t>(), n, ny, common.as<double>(), common.p, nrm_round_up(ny, 8), st));
	}
	NRM_TRY(nrm_de_sparse(d_y, y_dtype, ny, n, n, common.as<double>(), ncu, d_dci, L.ell.as<int16_t>(), L.ellv.as<double>(), L.base.as<int64_t>(), L.w.as<int32_t>(),
						  L.sig.as<int32_t>(), L.ngroups, L.slot2x.as<int32_t>(), d_bx, ldb, d_dot, ldd, by_gene, d_ssy, d_coefy, d_flags, d_c, n, ci, cval,
						  fused ? ct.as<double>() : nullptr, st));
	NRM_HIP(hipStreamSynchronize(st));  // (the scratch returns to the pool with this scope)
	return NRM_OK;
}

}  // namespace

int nrm_host_de_sparse(const void* d_x, int x_dtype, int64_t nx, const void* d_y, int y_dtype, int64_t ny, const double* d_c, const double* h_c64, int64_t nc, int64_t n,
					   const double* d_dci, int rank, double dof, int stat_kind, void* h_p, void* h_stat, void* h_alpha, void* h_varx, void* h_vary, void* h_r, void* h_t,
					   int out_dtype, int* taken, int64_t* handed_back) {
	hipStream_t st = nullptr;
	*taken = 0;
	*handed_back = 0;
	NrmDesignLists L;
	NRM_TRY(L.build(d_x, x_dtype, nx, n, true, 1.0 / 16, st));
	if (!L.ok) return NRM_OK;
	*taken = 1;
	const int64_t ncu = (rank > 0 && nc > 0) ? nc : 0;  // (covariates of rank 0 -- all zero -- leave the rows as they are: association.py:899-903)
	double cval = 0.0;
	const int ci = ncu ? constant_row(h_c64, nc, n, &cval) : -1;
	const int64_t nxp = nrm_round_up(nx, NRM_ROW_TILE), nyp = nrm_round_up(ny, NRM_ROW_TILE);
	DevBuf flags, ssx, bx, ssy, by, dot, op, ostat, orr, ot, oalpha;
	NRM_TRY(flags.alloc(16));
	NRM_HIP(hipMemsetAsync(flags.p, 0, 16, st));
	NRM_TRY(ssx.alloc((size_t)nxp * 8));
	NRM_HIP(hipMemsetAsync(ssx.p, 0, (size_t)nxp * 8, st));
	NRM_TRY(bx.alloc((size_t)nx * (nc > 0 ? nc : 1) * 8));
	NRM_HIP(hipMemsetAsync(bx.p, 0, (size_t)nx * (nc > 0 ? nc : 1) * 8, st));
	NRM_TRY(nrm_design_stats(L.row_ptr.as<int64_t>(), L.cells.as<int32_t>(), L.row_vals.as<double>(), ncu ? d_c : nullptr, n, ncu, ncu ? d_dci : nullptr, nx, ssx.as<double>(),
							 ncu ? bx.as<double>() : nullptr, flags.as<int32_t>(), st));
	NRM_TRY(ssy.alloc((size_t)nyp * 8));
	const bool want_alpha = h_alpha != nullptr && nc > 0;
	if (want_alpha) {
		NRM_TRY(by.alloc((size_t)ny * nc * 8));
		NRM_HIP(hipMemsetAsync(by.p, 0, (size_t)ny * nc * 8, st));
	}
	NRM_TRY(dot.alloc((size_t)nxp * nyp * 8));
	NRM_TRY(sparse_products(L, d_y, y_dtype, ny, n, d_c, ncu, ci, cval, d_dci, bx.as<double>(), nc > 0 ? nc : 1, dot.as<double>(), nyp, 0, ssy.as<double>(),
							(want_alpha && ncu) ? by.as<double>() : nullptr, flags.as<int32_t>(), st));
	const size_t ob = (size_t)nx * ny * nrm_esize(out_dtype);
	// the caller's result arrays are page-locked in place by a helper thread while the kernels run
	NrmHostPin pin_p, pin_s, pin_r, pin_t;
	std::thread pinner([&] {
		pin_p.try_pin(h_p, (int64_t)ob);
		pin_s.try_pin(h_stat, (int64_t)ob);
		pin_r.try_pin(h_r, (int64_t)ob);
		pin_t.try_pin(h_t, (int64_t)ob);
	});
	struct Join {
		std::thread& t;
		~Join() {
			if (t.joinable()) t.join();
		}
	} join{pinner};
	NRM_TRY(op.alloc(ob));
	NRM_TRY(ostat.alloc(ob));
	if (h_r) NRM_TRY(orr.alloc(ob));
	if (h_t) NRM_TRY(ot.alloc(ob));
	NRM_TRY(nrm_assoc_sweep(dot.as<double>(), nyp, ssx.as<double>(), ssy.as<double>(), nx, ny, n, dof, 0, stat_kind, op.p, ostat.p, h_r ? orr.p : nullptr, h_t ? ot.p : nullptr,
							out_dtype, ny, flags.as<int32_t>(), 0, nullptr, nullptr, 0.0, st));
	if (want_alpha) {
		NRM_TRY(oalpha.alloc(ob * nc));
		NRM_TRY(nrm_alpha(ostat.p, out_dtype, ny, stat_kind, ssx.as<double>(), n, bx.as<double>(), by.as<double>(), nx, ny, nc, oalpha.p, out_dtype, st));
	}
	int32_t hf[4];
	NRM_HIP(hipMemcpyAsync(hf, flags.p, 16, hipMemcpyDeviceToHost, st));
	NRM_HIP(hipStreamSynchronize(st));
	if (hf[0] || hf[1]) {
		nrm_set_error("association results failed the reference's assertions (association.py:248,252): %d tiles non-finite, %d tiles with R^2 > 1+1e-8", hf[0], hf[1]);
		return NRM_E_NUMERIC;
	}
	if (hf[2] > 0) {  // rows all but inside the span of the covariates: the caller redoes the call on K1 and the fp64 Gram kernel
		*handed_back = hf[2];
		return NRM_OK;
	}
	pinner.join();
	NRM_HIP(hipMemcpy(h_p, op.p, ob, hipMemcpyDeviceToHost));
	NRM_HIP(hipMemcpy(h_stat, ostat.p, ob, hipMemcpyDeviceToHost));
	if (h_r) NRM_HIP(hipMemcpy(h_r, orr.p, ob, hipMemcpyDeviceToHost));
	if (h_t) NRM_HIP(hipMemcpy(h_t, ot.p, ob, hipMemcpyDeviceToHost));
	if (want_alpha) NRM_HIP(hipMemcpy(h_alpha, oalpha.p, ob * nc, hipMemcpyDeviceToHost));
	NRM_TRY(emit_var(ssy.as<double>(), ny, n, h_vary, out_dtype));
	if (h_varx) NRM_TRY(emit_var(ssx.as<double>(), nx, n, h_varx, out_dtype));
	return NRM_OK;
}

// ---- de with nx + nc <= 32 (case-control DE: BASELINE configs[2]) -------------------------------------------------------------------------------------
// The raw expression rows streamed once against Z = [C; X~] on the fp64 matrix cores (csrc/nrm_gram_skinny.hip), never residualised, never
// quantised: what engine.association_de_streaming does for the Python host, kernel for kernel.  A constant covariate (the intercept) leaves Z -- its
// product with a row is a plain sum the kernel takes on the vector ALU -- by moving to the end of the covariates; dci is permuted with it (no rank
// assumption) and alpha is put back in the caller's order.  d_x: the design rows already in HBM (pitch n); h_dy: uploaded here, into rows zero padded
// to 16 cells when n is not a multiple of 16 (the kernel streams 16-cell slabs without bounds checks).
int nrm_host_de_streaming(const void* d_x, int x_dtype, int64_t nx, const void* h_dy, int y_dtype, int64_t ny, const double* h_c64, int64_t nc, int64_t n,
						  const double* h_dci, int rank, double dof, int stat_kind, void* h_p, void* h_stat, void* h_alpha, void* h_varx, void* h_vary, void* h_r, void* h_t,
						  int out_dtype) {
	hipStream_t st = nullptr;
	NRM_REQUIRE(nx + nc <= 32 && nx > 0 && ny > 0, "nrm_host_de_streaming: needs nx + nc <= 32");
	double cval = 0.0;
	int ci = nc ? constant_row(h_c64, nc, n, &cval) : -1;
	if (nc + nx > 31 + (ci >= 0 ? 1 : 0)) {
		ci = -1;
		cval = 0.0;
	}
	const int const_last = ci >= 0 ? 1 : 0;
	const int64_t ncz = nc - const_last;  // covariate rows that stay in Z
	std::vector<int64_t> perm((size_t)nc);
	{
		int64_t j = 0;
		for (int64_t c = 0; c < nc; c++)
			if (c != ci) perm[(size_t)j++] = c;
		if (ci >= 0) perm[(size_t)j] = ci;
	}
	DevBuf cz, dciz, z, xpad, gx, xwork, rwork, ssx, coefx, ypad, yraw, g, ssraw, swork, ssy, by, flags, op, ostat, orr, ot, oalpha;
	if (nc) {  // the covariates in Z's order (the constant one last) and their pseudo-inverse permuted with them
		std::vector<double> hc((size_t)nc * n), hd((size_t)nc * nc);
		for (int64_t c = 0; c < nc; c++) memcpy(&hc[(size_t)(c * n)], h_c64 + perm[(size_t)c] * n, (size_t)n * 8);
		for (int64_t a = 0; a < nc; a++)
			for (int64_t b = 0; b < nc; b++) hd[(size_t)(a * nc + b)] = h_dci ? h_dci[perm[(size_t)a] * nc + perm[(size_t)b]] : 0.0;
		NRM_TRY(cz.alloc(hc.size() * 8));
		NRM_HIP(hipMemcpy(cz.p, hc.data(), hc.size() * 8, hipMemcpyHostToDevice));
		NRM_TRY(dciz.alloc(hd.size() * 8));
		NRM_HIP(hipMemcpy(dciz.p, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
	}
	const int64_t k32 = nrm_round_up(n, 128), n16 = nrm_round_up(n, 16);
	NRM_TRY(z.alloc((size_t)32 * k32 * 8));
	NRM_HIP(hipMemsetAsync(z.p, 0, (size_t)32 * k32 * 8, st));
	if (ncz) NRM_TRY(nrm_copy_rows(z.p, k32 * 8, cz.p, n * 8, n * 8, ncz, st));
	// the design rows: readable up to a multiple of 16 cells
	const size_t xe = nrm_esize(x_dtype), ye = nrm_esize(y_dtype);
	const void* xd = d_x;
	int64_t ldx = n;
	if (n16 != n) {
		NRM_TRY(xpad.alloc((size_t)nx * n16 * xe));
		NRM_HIP(hipMemsetAsync(xpad.p, 0, (size_t)nx * n16 * xe, st));
		NRM_TRY(nrm_copy_rows(xpad.p, n16 * xe, d_x, n * xe, n * xe, nx, st));
		xd = xpad.p;
		ldx = n16;
	}
	NRM_TRY(gx.alloc((size_t)256 * 32 * 8));
	NRM_HIP(hipMemsetAsync(gx.p, 0, (size_t)256 * 32 * 8, st));
	const bool active = rank > 0 && nc > 0;
	if (active) {  // a = x C^T against the covariates in Z's order (the constant one: column 31)
		NRM_TRY(xwork.alloc((size_t)nrm_design_products_workspace_doubles(nx, n) * 8));
		NRM_TRY(nrm_design_products(xd, x_dtype, nx, n, ldx, cz.as<double>(), nc, n, gx.as<double>(), xwork.as<double>(), const_last, st));
	}
	const bool want_alpha = h_alpha != nullptr && nc > 0;
	NRM_TRY(rwork.alloc((size_t)64 * ((k32 + 1023) / 1024) * 8));
	NRM_TRY(ssx.alloc((size_t)NRM_ROW_TILE * 8));
	NRM_HIP(hipMemsetAsync(ssx.p, 0, (size_t)NRM_ROW_TILE * 8, st));
	if (want_alpha) {
		NRM_TRY(coefx.alloc((size_t)nx * nc * 8));
		NRM_HIP(hipMemsetAsync(coefx.p, 0, (size_t)nx * nc * 8, st));
	}
	double* xt = z.as<double>() + ncz * k32;  // the residualised design rows go straight into their rows of Z
	NRM_TRY(nrm_residualize_wide(xd, x_dtype, nx, n, ldx, nc ? cz.as<double>() : nullptr, nc, n, gx.as<double>(), nc ? dciz.as<double>() : nullptr, rank, xt, k32, ssx.as<double>(),
								 want_alpha ? coefx.as<double>() : nullptr, rwork.as<double>(), const_last, st));
	// the expression rows
	const void* yd;
	int64_t ldy = n;
	if (n16 == n) {
		NRM_TRY(ypad.alloc((size_t)ny * n * ye));
		NRM_TRY(nrm_upload(h_dy, ypad.p, (int64_t)ny * n * ye, 0, (void*)st));
		yd = ypad.p;
	} else {
		NRM_TRY(yraw.alloc((size_t)ny * n * ye));
		NRM_TRY(nrm_upload(h_dy, yraw.p, (int64_t)ny * n * ye, 0, (void*)st));
		NRM_TRY(ypad.alloc((size_t)ny * n16 * ye));
		NRM_HIP(hipMemsetAsync(ypad.p, 0, (size_t)ny * n16 * ye, st));
		NRM_TRY(nrm_copy_rows(ypad.p, n16 * ye, yraw.p, n * ye, n * ye, ny, st));
		yraw.release();
		yd = ypad.p;
		ldy = n16;
	}
	const int64_t nyp = nrm_round_up(ny, 256);
	NRM_TRY(g.alloc((size_t)nyp * 32 * 8));
	NRM_TRY(ssraw.alloc((size_t)nyp * 8));
	NRM_TRY(swork.alloc((size_t)nrm_gram_skinny_workspace_bytes()));
	NRM_TRY(nrm_gram_skinny(yd, y_dtype, ny, n, ldy, z.as<double>(), k32, k32, g.as<double>(), ssraw.as<double>(), nyp, ncz + nx, cval, swork.p, st));
	const size_t ob = (size_t)nx * ny * nrm_esize(out_dtype);
	NRM_TRY(op.alloc(ob));
	NRM_TRY(ostat.alloc(ob));
	if (h_r) NRM_TRY(orr.alloc(ob));
	if (h_t) NRM_TRY(ot.alloc(ob));
	NRM_TRY(ssy.alloc((size_t)nyp * 8));
	if (want_alpha) {
		NRM_TRY(by.alloc((size_t)ny * nc * 8));
		NRM_HIP(hipMemsetAsync(by.p, 0, (size_t)ny * nc * 8, st));
	}
	NRM_TRY(flags.alloc(16));
	NRM_HIP(hipMemsetAsync(flags.p, 0, 16, st));
	NRM_TRY(nrm_de_small_sweep(g.as<double>(), ssraw.as<double>(), nc ? dciz.as<double>() : nullptr, nc, rank, ssx.as<double>(), nx, ny, n, dof, stat_kind, op.p, ostat.p,
							   h_r ? orr.p : nullptr, h_t ? ot.p : nullptr, out_dtype, ny, ssy.as<double>(), want_alpha ? by.as<double>() : nullptr, flags.as<int32_t>(), const_last, st));
	if (want_alpha) {
		NRM_TRY(oalpha.alloc(ob * nc));
		NRM_TRY(nrm_alpha(ostat.p, out_dtype, ny, stat_kind, ssx.as<double>(), n, coefx.as<double>(), by.as<double>(), nx, ny, nc, oalpha.p, out_dtype, st));
	}
	int32_t hf[4];
	NRM_HIP(hipMemcpyAsync(hf, flags.p, 16, hipMemcpyDeviceToHost, st));
	NRM_HIP(hipStreamSynchronize(st));
	if (hf[0] || hf[1]) {
		nrm_set_error("association results failed the reference's assertions (association.py:248,252): %d tiles non-finite, %d tiles with R^2 > 1+1e-8", hf[0], hf[1]);
		return NRM_E_NUMERIC;
	}
	NRM_HIP(hipMemcpy(h_p, op.p, ob, hipMemcpyDeviceToHost));
	NRM_HIP(hipMemcpy(h_stat, ostat.p, ob, hipMemcpyDeviceToHost));
	if (h_r) NRM_HIP(hipMemcpy(h_r, orr.p, ob, hipMemcpyDeviceToHost));
	if (h_t) NRM_HIP(hipMemcpy(h_t, ot.p, ob, hipMemcpyDeviceToHost));
	if (want_alpha) {  // the coefficients came out in Z's covariate order: back to the caller's
		const size_t es = nrm_esize(out_dtype), cnt = (size_t)nx * ny;
		std::vector<char> tmp(ob * nc);
		NRM_HIP(hipMemcpy(tmp.data(), oalpha.p, ob * nc, hipMemcpyDeviceToHost));
		for (size_t i = 0; i < cnt; i++)
			for (int64_t c = 0; c < nc; c++) memcpy((char*)h_alpha + (i * nc + (size_t)perm[(size_t)c]) * es, tmp.data() + (i * nc + (size_t)c) * es, es);
	}
	NRM_TRY(emit_var(ssy.as<double>(), ny, n, h_vary, out_dtype));
	if (h_varx) NRM_TRY(emit_var(ssx.as<double>(), nx, n, h_varx, out_dtype));
	return NRM_OK;
}

// ---- shared pieces of the entries below ---------------------------------------------------------------------------------------------------
namespace {

// host covariates as fp64 (nc, n), on the host and on the device
int covariates_f64(const void* h_dc, int c_dtype, int64_t nc, int64_t n, std::vector<double>& c64, DevBuf& dc) {
	if (nc <= 0) return NRM_OK;
	c64.resize((size_t)nc * n);
	if (c_dtype == NRM_F64)
		memcpy(c64.data(), h_dc, c64.size() * 8);
	else
		for (size_t i = 0; i < c64.size(); i++) c64[i] = ((const float*)h_dc)[i];
	NRM_TRY(dc.alloc(c64.size() * 8));
	NRM_HIP(hipMemcpy(dc.p, c64.data(), c64.size() * 8, hipMemcpyHostToDevice));
	return NRM_OK;
}

int upload_matrix(const void* h, int dtype, int64_t rows, int64_t n, DevBuf& d, hipStream_t st) {
	NRM_TRY(d.alloc((size_t)rows * n * nrm_esize(dtype)));
	return nrm_upload(h, d.p, rows * n * (int64_t)nrm_esize(dtype), 0, (void*)st);
}

template <typename T>
int download(std::vector<T>& h, const void* d, size_t count) {
	h.resize(count);
	NRM_HIP(hipMemcpy(h.data(), d, count * sizeof(T), hipMemcpyDeviceToHost));
	return NRM_OK;
}

int check_flags2(const int32_t* d_flags, hipStream_t st) {
	int32_t hf[2];
	NRM_HIP(hipMemcpyAsync(hf, d_flags, 8, hipMemcpyDeviceToHost, st));
	NRM_HIP(hipStreamSynchronize(st));
	if (hf[0] || hf[1]) {
		nrm_set_error("association results failed the reference's assertions (association.py:248,252): %d non-finite, %d with R^2 > 1+1e-8", hf[0], hf[1]);
		return NRM_E_NUMERIC;
	}
	return NRM_OK;
}

// results (rows x cols of out_dtype) device -> the caller's array, page-locked for the copy when it is large
int copy_out(void* h, const void* d, size_t bytes) {
	if (!h || !bytes) return NRM_OK;
	NrmHostPin pin;
	pin.try_pin(h, (int64_t)bytes);
	NRM_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
	return NRM_OK;
}

}  // namespace

// ---- single=1 (association.py:263-390,911-925) for a design with entries >= 0 ----------------------------------------------------------------
extern "C" int nrm_association_tests_single1_host(const void* h_dx, int x_dtype, int64_t nx, const void* h_dy, int y_dtype, int64_t ny, const void* h_dc, int c_dtype,
												   int64_t nc, int64_t n, int dimreduce, int return_dot, void* h_p, void* h_stat, void* h_alpha, void* h_varx,
												   void* h_vary, int out_dtype) {
	std::lock_guard<std::mutex> serial(nrm_host_entry_mutex());
	NRM_TRY(nrm_bind_device());
	NRM_REQUIRE(h_dx && h_dy && nx > 0 && ny > 0 && n > 0 && nc >= 0 && (nc == 0 || h_dc), "Unmatching dx/dy/dc dimensions.");
	NRM_REQUIRE(h_p && h_stat && h_varx && h_vary, "nrm_association_tests_single1_host: null output");
	NRM_REQUIRE((x_dtype == NRM_F32 || x_dtype == NRM_F64) && (y_dtype == NRM_F32 || y_dtype == NRM_F64) && (out_dtype == NRM_F32 || out_dtype == NRM_F64), "bad dtype");
	if (nc > 32) {
		nrm_set_error("nrm_association_tests_single1_host covers up to 32 covariates (the package's masked-Gram path takes more)");
		return NRM_E_UNSUPPORTED;
	}
	hipStream_t st = nullptr;
	DevBuf dx, dy, dc;
	std::vector<double> c64;
	NRM_TRY(upload_matrix(h_dx, x_dtype, nx, n, dx, st));
	NRM_TRY(covariates_f64(h_dc, c_dtype, nc, n, c64, dc));
	// the design's entries (CSR) and what they are like: dx.max() == 1 (association.py:914)
	NrmDesignLists L;
	NRM_TRY(L.build(dx.p, x_dtype, nx, n, false, 0.25, st));
	if (!(L.bits & NRM_DESIGN_HAS1) || (L.bits & (NRM_DESIGN_GT1 | NRM_DESIGN_NAN))) {
		nrm_set_error("the largest entry of dx must be 1 (association.py:914)");
		return NRM_E_NUMERIC;
	}
	if (!L.ok || (L.bits & NRM_DESIGN_NEG)) {
		nrm_set_error("nrm_association_tests_single1_host covers designs with entries >= 0 of which at most a quarter are set (the package's masked-Gram path takes the others)");
		return NRM_E_UNSUPPORTED;
	}
	// cell selection (association.py:915-918)
	const int64_t nnz = L.nnz;
	const int64_t gb = nrm_single1_select_gram_blocks(), nb = (nc + 7) / 8, npairs = nb * (nb + 1) / 2;
	DevBuf cnt, code, seg, idx, xe, ce, rowinfo, gpart, info;
	NRM_TRY(cnt.alloc((size_t)n * 4));
	NRM_TRY(code.alloc((size_t)n * 4));
	NRM_TRY(seg.alloc((size_t)(nx + 1) * 8));
	NRM_TRY(idx.alloc((size_t)nnz * 8));
	NRM_TRY(xe.alloc((size_t)nnz * 8));
	if (nc) NRM_TRY(ce.alloc((size_t)nnz * nc * 8));
	NRM_TRY(rowinfo.alloc((size_t)nx * 3 * 8));
	if (nc) NRM_TRY(gpart.alloc((size_t)npairs * gb * 64 * 8));
	NRM_TRY(info.alloc(64));
	NRM_TRY(nrm_single1_select(L.row_ptr.as<int64_t>(), L.cells.as<int32_t>(), L.row_vals.as<double>(), nx, n, nnz, dc.as<double>(), n, nc, cnt.as<int32_t>(),
							   code.as<int32_t>(), seg.as<int64_t>(), idx.as<int64_t>(), xe.as<double>(), ce.as<double>(), rowinfo.as<double>(), gpart.as<double>(),
							   info.as<int64_t>(), st));
	std::vector<int64_t> hinfo;
	NRM_TRY(download(hinfo, info.p, 8));  // (the one read-back in front of the stream kernel: the size of its transposed output)
	const int64_t n_common = hinfo[3], n_e = hinfo[4];
	const int64_t pitch = 26 + nc + nc * nc, npair = nc * (nc + 1) / 2, gsw = npair + nc + 1;
	const int64_t ldye = nrm_round_up(ny, 8);
	const size_t ob = (size_t)nx * ny * nrm_esize(out_dtype);
	DevBuf ye, common, drec, dvarx, op, ostat, ovary, oalpha, flags;
	NRM_TRY(drec.alloc((size_t)nx * pitch * 8));
	NRM_TRY(dvarx.alloc((size_t)nx * 8));
	NRM_TRY(flags.alloc(32));
	NRM_HIP(hipMemsetAsync(flags.p, 0, 32, st));
	std::vector<double> vxx((size_t)nx);
	if (nc <= 8) {
		// Round 6: the groupings' statistics -- M_i = C_S C_S^T, its pseudo-inverse and integer rank, ccx, vx, dof, the P-value plan (association.py:350-374) --
		// stay on the device (nrm_single1_group_stats + nrm_single1_group_info, the launches of single1.Single1Plan): nothing comes back but the counters and varx
		DevBuf gsd;
		NRM_TRY(gsd.alloc((size_t)nx * gsw * 8));
		NRM_TRY(nrm_single1_group_stats(seg.as<int64_t>(), idx.as<int64_t>(), xe.as<double>(), dc.as<double>(), n, nc, nx, gsd.as<double>(), st));
		NRM_TRY(nrm_single1_group_info(gsd.as<double>(), gpart.as<double>(), rowinfo.as<double>(), info.as<int64_t>(), nc, nx, dimreduce, drec.as<double>(), pitch,
									   dvarx.as<double>(), flags.as<int32_t>(), st));
		NRM_HIP(hipStreamSynchronize(st));  // (gsd is released at the end of this block)
	} else {
		// more than 8 covariates: the statistics on the host, as in rounds 4-5
		std::vector<double> hrows, hpart;
		NRM_TRY(download(hrows, rowinfo.p, (size_t)nx * 3));
		std::vector<double> ns((size_t)nx);
		for (int64_t i = 0; i < nx; i++) {
			ns[(size_t)i] = (double)n_common + hrows[(size_t)i * 3];
			double lo = hrows[(size_t)i * 3 + 1], hi = hrows[(size_t)i * 3 + 2];
			if (n_common > 0) {
				lo = lo < 0.0 ? lo : 0.0;
				hi = hi > 0.0 ? hi : 0.0;
			}
			if (!(hi > lo)) {  // > 1 distinct value among the selected cells (:917-918)
				nrm_set_error("grouping %lld has a single value on the cells selected for it (association.py:917-918)", (long long)i);
				return NRM_E_NUMERIC;
			}
		}
		std::vector<double> mcc((size_t)nc * nc, 0.0);
		{  // covariate Gram of the shared cells: the kernel's partial sums added up in a fixed order
			NRM_TRY(download(hpart, gpart.p, (size_t)npairs * gb * 64));
			int64_t q = 0;
			for (int64_t bi = 0; bi < nb; bi++)
				for (int64_t bj = bi; bj < nb; bj++, q++) {
					double blk[64];
					for (int e = 0; e < 64; e++) blk[e] = 0.0;
					for (int64_t g = 0; g < gb; g++)
						for (int e = 0; e < 64; e++) blk[e] += hpart[(size_t)((q * gb + g) * 64 + e)];
					for (int i = 0; i < 8; i++)
						for (int j = 0; j < 8; j++) {
							const int64_t a = bi * 8 + i, b = bj * 8 + j;
							if (a < nc && b < nc) mcc[(size_t)(a * nc + b)] = mcc[(size_t)(b * nc + a)] = blk[i * 8 + j];
						}
				}
		}
		// the groupings' own sums over their own cells: M_i = C_S C_S^T, C_S x_S, |x_S|^2 (association.py:350-364)
		std::vector<double> gs((size_t)nx * gsw, 0.0);
		{
			std::vector<int64_t> hseg, hidx;
			std::vector<double> hxe;
			NRM_TRY(download(hseg, seg.p, (size_t)nx + 1));
			NRM_TRY(download(hidx, idx.p, (size_t)n_e));
			NRM_TRY(download(hxe, xe.p, (size_t)n_e));
			for (int64_t i = 0; i < nx; i++) {
				double* o = &gs[(size_t)i * gsw];
				for (int64_t e = hseg[(size_t)i]; e < hseg[(size_t)i + 1]; e++) {
					const int64_t k = hidx[(size_t)e];
					const double x = hxe[(size_t)e];
					int64_t w = 0;
					for (int64_t c = 0; c < nc; c++)
						for (int64_t d = c; d < nc; d++) o[w++] += c64[(size_t)(c * n + k)] * c64[(size_t)(d * n + k)];
					for (int64_t c = 0; c < nc; c++) o[w++] += c64[(size_t)(c * n + k)] * x;
					o[w] += x * x;
				}
			}
		}
		// per grouping: pseudo-inverse of M_i (integer rank), ccx, vx, dof, the P-value plan
		std::vector<double> rec((size_t)nx * pitch, 0.0), dof((size_t)nx);
		std::vector<int64_t> rk((size_t)nx, 0);
		std::vector<double> mc((size_t)nx * nc * nc), mi((size_t)nx * nc * nc);
		for (int64_t i = 0; i < nx; i++) {
			const double* o = &gs[(size_t)i * gsw];
			int64_t w = 0;
			for (int64_t c = 0; c < nc; c++)
				for (int64_t d = c; d < nc; d++, w++)
					mc[(size_t)((i * nc + c) * nc + d)] = mc[(size_t)((i * nc + d) * nc + c)] = o[w] + mcc[(size_t)(c * nc + d)];
		}
		for (size_t e = 0; e < mc.size(); e++)
			if (!std::isfinite(mc[e])) {
				nrm_set_error("array must not contain infs or NaNs");
				return NRM_E_ARG;
			}
		NRM_TRY(nrm_small_pinv(mc.data(), nx, nc, 1e-8, mi.data(), rk.data(), 0));  // association.py:350-351
		for (int64_t i = 0; i < nx; i++) {
			double* r = &rec[(size_t)i * pitch];
			const double* o = &gs[(size_t)i * gsw];
			const double* xc = o + npair;
			double* m = &mi[(size_t)i * nc * nc];
			if (rk[(size_t)i] == 0) memset(m, 0, (size_t)nc * nc * 8);
			double xx = o[npair + nc];
			for (int64_t c = 0; c < nc; c++) {
				double t = 0.0;
				for (int64_t d = 0; d < nc; d++) t += m[c * nc + d] * xc[d];
				r[26 + c] = t;  // ccx
			}
			for (int64_t c = 0; c < nc; c++) xx -= xc[c] * r[26 + c];
			memcpy(r + 26 + nc, m, (size_t)nc * nc * 8);
			vxx[(size_t)i] = xx / ns[(size_t)i];
		}
		for (int64_t i = 0; i < nx; i++) {
			if (vxx[(size_t)i] == 0.0) vxx[(size_t)i] = 1.0;  // association.py:362-364
			dof[(size_t)i] = ns[(size_t)i] - 1 - (double)rk[(size_t)i] - dimreduce;
			if (dof[(size_t)i] <= 0) {
				nrm_set_error("Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.");
				return NRM_E_DEVICE;
			}
			rec[(size_t)i * pitch] = ns[(size_t)i];
			rec[(size_t)i * pitch + 1] = vxx[(size_t)i];
		}
		NRM_TRY(nrm_pvalue_plan_init_many(dof.data(), nx, rec.data() + 2, pitch));
		NRM_HIP(hipMemcpy(drec.p, rec.data(), rec.size() * 8, hipMemcpyHostToDevice));
		NRM_HIP(hipMemcpy(dvarx.p, vxx.data(), vxx.size() * 8, hipMemcpyHostToDevice));
	}
	// the expression matrix, read once where it lies: sums over the shared cells, the values at the groupings' own cells transposed
	NRM_TRY(upload_matrix(h_dy, y_dtype, ny, n, dy, st));
	NRM_TRY(ye.alloc((size_t)(n_e > 0 ? n_e : 1) * ldye * nrm_esize(y_dtype) + 64));
	NRM_TRY(common.alloc((size_t)(nc + 1) * ny * 8));
	NRM_TRY(nrm_single1_stream(dy.p, y_dtype, n, dc.as<double>(), n, nc, code.as<int32_t>(), n, ny, common.as<double>(), ye.p, ldye, st));
	NRM_TRY(op.alloc(ob));
	NRM_TRY(ostat.alloc(ob));
	NRM_TRY(ovary.alloc(ob));
	if (h_alpha && nc) {
		NRM_TRY(oalpha.alloc(ob * nc));
		NRM_HIP(hipMemsetAsync(oalpha.p, 0, ob * nc, st));
	}
	NRM_TRY(nrm_single1_cells(ye.p, y_dtype, ldye, ce.as<double>(), xe.as<double>(), seg.as<int64_t>(), common.as<double>(), drec.as<double>(), pitch, nc, nx, ny, return_dot,
							  op.p, ostat.p, ovary.p, (h_alpha && nc) ? oalpha.p : nullptr, out_dtype, ny, flags.as<int32_t>(), st));
	{  // what the reference asserts or raises, in its order: the selection (:917-918), the SVD's finiteness check, the cell count, the results (:248,252)
		int32_t hf[8];
		NRM_HIP(hipMemcpyAsync(hf, flags.p, 32, hipMemcpyDeviceToHost, st));
		NRM_HIP(hipStreamSynchronize(st));
		if (hf[2]) {
			nrm_set_error("%d groupings take a single value on the cells selected for them (association.py:917-918)", hf[2]);
			return NRM_E_NUMERIC;
		}
		if (hf[4]) {
			nrm_set_error("array must not contain infs or NaNs");
			return NRM_E_ARG;
		}
		if (hf[3]) {
			nrm_set_error("Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.");
			return NRM_E_DEVICE;
		}
		if (hf[0] || hf[1]) {
			nrm_set_error("association results failed the reference's assertions (association.py:248,252): %d non-finite, %d with R^2 > 1+1e-8", hf[0], hf[1]);
			return NRM_E_NUMERIC;
		}
	}
	NRM_TRY(copy_out(h_p, op.p, ob));
	NRM_TRY(copy_out(h_stat, ostat.p, ob));
	NRM_TRY(copy_out(h_vary, ovary.p, ob));
	if (h_alpha && nc) NRM_TRY(copy_out(h_alpha, oalpha.p, ob * nc));
	NRM_HIP(hipMemcpy(vxx.data(), dvarx.p, (size_t)nx * 8, hipMemcpyDeviceToHost));
	for (int64_t i = 0; i < nx; i++) {
		if (out_dtype == NRM_F64)
			((double*)h_varx)[i] = vxx[(size_t)i];
		else
			((float*)h_varx)[i] = (float)vxx[(size_t)i];
	}
	return NRM_OK;
}

// ---- the two pieces the package needs to follow the reference's per-grouping algorithm of single=4 WITHOUT torch ------------------------------------------
// (association.py:926-980 forms the Gram matrices of A = [dx; dc] and of dy against A with numpy.matmul and hands them to association_test_4; for designs the closed
// form does not cover -- rank-deficient A A^T, mpc / method / qr, dy=None -- nrm_association_tests_single4_host answers NRM_E_UNSUPPORTED and normalisr_amd/single4.py
// runs that algorithm on these products: the contractions on the fp64 Gram kernel, the small pseudo-inverses in numpy, the P-values by nrm_pvalues_host.)
// h_out (ra, rb) fp64 = A B^T over n cells; h_b == NULL: B = A (rb = ra).  h_ssa (ra) / h_ssb (rb) or NULL: the rows' sums of squares.
extern "C" int nrm_gram_host(const void* h_a, int a_dtype, int64_t ra, const void* h_b, int b_dtype, int64_t rb, int64_t n, double* h_out, double* h_ssa, double* h_ssb) {
	std::lock_guard<std::mutex> serial(nrm_host_entry_mutex());
	NRM_TRY(nrm_bind_device());
	NRM_REQUIRE(h_a && h_out && ra > 0 && n > 0 && (a_dtype == NRM_F32 || a_dtype == NRM_F64) && (!h_b || (rb > 0 && (b_dtype == NRM_F32 || b_dtype == NRM_F64))),
				"nrm_gram_host: bad arguments");
	if (!h_b) rb = ra;
	hipStream_t st = nullptr;
	const int64_t kp = nrm_round_up(n, NRM_K_TILE), rap = nrm_round_up(ra, NRM_ROW_TILE), rbp = nrm_round_up(rb, NRM_ROW_TILE);
	DevBuf raw, a64, b64, ssa, ssb, dot, gwork;
	auto padded = [&](const void* h, int dtype, int64_t rows, int64_t rp, DevBuf& out, DevBuf& ss) -> int {
		NRM_TRY(upload_matrix(h, dtype, rows, n, raw, st));
		NRM_TRY(out.alloc((size_t)rp * kp * 8));
		NRM_TRY(ss.alloc((size_t)rp * 8));
		NRM_TRY(nrm_residualize(raw.p, dtype, rows, n, n, nullptr, 0, 0, nullptr, 0, out.as<double>(), kp, rp, ss.as<double>(), nullptr, st));  // no covariates: the fp64 padded copy + sums of squares
		NRM_HIP(hipStreamSynchronize(st));
		raw.release();
		return NRM_OK;
	};
	NRM_TRY(padded(h_a, a_dtype, ra, rap, a64, ssa));
	if (h_b) NRM_TRY(padded(h_b, b_dtype, rb, rbp, b64, ssb));
	NRM_TRY(dot.alloc((size_t)rap * rbp * 8));
	NRM_HIP(hipMemsetAsync(dot.p, 0, (size_t)rap * rbp * 8, st));  // (the kernel leaves pure-padding sub-blocks unwritten)
	NRM_TRY(gwork.alloc((size_t)nrm_gram_workspace_bytes()));
	NRM_TRY(nrm_gram_f64(a64.as<double>(), h_b ? b64.as<double>() : a64.as<double>(), rap, rbp, kp, kp, kp, dot.as<double>(), rbp, 0, ra, rb, gwork.p, st));
	std::vector<double> hd;
	NRM_TRY(download(hd, dot.p, (size_t)rap * rbp));
	for (int64_t i = 0; i < ra; i++) memcpy(h_out + i * rb, &hd[(size_t)(i * rbp)], (size_t)rb * 8);
	if (h_ssa) NRM_HIP(hipMemcpy(h_ssa, ssa.p, (size_t)ra * 8, hipMemcpyDeviceToHost));
	if (h_ssb) NRM_HIP(hipMemcpy(h_ssb, h_b ? ssb.p : ssa.p, (size_t)rb * 8, hipMemcpyDeviceToHost));
	return NRM_OK;
}

// h_p[i] = I_{1 - h_r2[i]}(dof / 2, 1 / 2): scipy.stats.beta.cdf(1 - R2, dof / 2, 0.5) of association.py:563 for host arrays (the device function of nrm_pvalue.h)
extern "C" int nrm_pvalues_host(const double* h_r2, int64_t count, double dof, double* h_p) {
	std::lock_guard<std::mutex> serial(nrm_host_entry_mutex());
	NRM_TRY(nrm_bind_device());
	NRM_REQUIRE(count >= 0 && (count == 0 || (h_r2 && h_p)), "nrm_pvalues_host: bad arguments");
	if (count == 0) return NRM_OK;
	hipStream_t st = nullptr;
	DevBuf r2, p;
	NRM_TRY(r2.alloc((size_t)count * 8));
	NRM_TRY(p.alloc((size_t)count * 8));
	NRM_HIP(hipMemcpy(r2.p, h_r2, (size_t)count * 8, hipMemcpyHostToDevice));
	NRM_TRY(nrm_pvalues_from_r2(r2.as<double>(), count, dof, p.as<double>(), st));
	NRM_HIP(hipMemcpy(h_p, p.p, (size_t)count * 8, hipMemcpyDeviceToHost));
	return NRM_OK;
}

// ---- binnet (binnet.py:134-173) ------------------------------------------------------------------------------------------------------------
extern "C" int nrm_binnet_host(const void* h_p, int p_dtype, int64_t ng, double qcut, unsigned char* h_net, int64_t* total) {
	std::lock_guard<std::mutex> serial(nrm_host_entry_mutex());
	NRM_TRY(nrm_bind_device());
	NRM_REQUIRE(h_p && h_net && total && ng > 0 && (p_dtype == NRM_F32 || p_dtype == NRM_F64), "nrm_binnet_host: bad arguments");
	NRM_REQUIRE(qcut > 0 && qcut < 1, "qcut must be between 0 and 1.");
	hipStream_t st = nullptr;
	DevBuf dp, dn, tot, flags;
	NRM_TRY(upload_matrix(h_p, p_dtype, ng, ng, dp, st));
	NRM_TRY(dn.alloc((size_t)ng * ng));
	NRM_TRY(tot.alloc(8));
	NRM_TRY(flags.alloc(16));
	NRM_HIP(hipMemsetAsync(tot.p, 0, 8, st));
	NRM_HIP(hipMemsetAsync(flags.p, 0, 16, st));
	NRM_TRY(nrm_binnet(dp.p, p_dtype, ng, ng, qcut, dn.as<unsigned char>(), ng, tot.as<unsigned long long>(), flags.as<int32_t>(), st));
	int32_t hf[4];
	NRM_HIP(hipMemcpyAsync(hf, flags.p, 16, hipMemcpyDeviceToHost, st));
	NRM_HIP(hipStreamSynchronize(st));
	if (hf[0]) {
		nrm_set_error("P-values must be finite and within [0, 1] (binnet.py:151-152): %d rows are not", hf[0]);
		return NRM_E_NUMERIC;
	}
	unsigned long long t = 0;
	NRM_HIP(hipMemcpy(&t, tot.p, 8, hipMemcpyDeviceToHost));
	*total = (int64_t)t;
	return copy_out(h_net, dn.p, (size_t)ng * ng);
}

// ---- normvar (norm.py:166-289): the expression side, whole problem ------------------------------------------------------------------------------
// h_y (rows, n) fp32 / fp64 -> h_out (rows, n): gene g multiplied by e_gk = w_k^wt_g, the covariates C e_g removed, the variance put back (keepvar).  The
// three kernels of csrc/nrm_normvar.hip as normalisr_amd.norm.normvar launches them: per-gene moments in one pass, a lane per gene solving its small OLS with the
// reference's rank rule, one pass writing the result.  1 .. nrm_normvar_device_covariates() covariates (NRM_E_UNSUPPORTED beyond: the package's Gram-launch form);
// *zero_rank = genes whose covariates have rank 0 (norm.py:158-159 raises for them: the caller does); non-finite results: NRM_E_NUMERIC (norm.py:286).
// The covariates' own scaling (norm.py:261-273) is a few numpy lines on (nc, n) and stays with the caller.
extern "C" int nrm_normvar_host(const void* h_y, int y_dtype, int64_t rows, int64_t n, const double* h_lnw, const double* h_wt, const double* h_c, int64_t nc, double tol,
								int keepvar, void* h_out, int out_dtype, int64_t* zero_rank) {
	std::lock_guard<std::mutex> serial(nrm_host_entry_mutex());
	NRM_TRY(nrm_bind_device());
	NRM_REQUIRE(h_y && h_lnw && h_wt && h_c && h_out && zero_rank && rows > 0 && n > 0 && nc > 0, "nrm_normvar_host: bad arguments");
	NRM_REQUIRE((y_dtype == NRM_F32 || y_dtype == NRM_F64) && (out_dtype == NRM_F32 || out_dtype == NRM_F64), "nrm_normvar_host: bad dtype");
	if (nc > 32) {
		nrm_set_error("nrm_normvar_host: at most 32 covariates (the package's Gram-launch form with numpy's stacked SVD takes more)");
		return NRM_E_UNSUPPORTED;
	}
	hipStream_t st = nullptr;
	DevBuf y, lnw, wt, c, mom, b, scale, rank, flags, out;
	NRM_TRY(upload_matrix(h_y, y_dtype, rows, n, y, st));
	NRM_TRY(upload_matrix(h_lnw, NRM_F64, 1, n, lnw, st));
	NRM_TRY(upload_matrix(h_wt, NRM_F64, 1, rows, wt, st));
	NRM_TRY(upload_matrix(h_c, NRM_F64, nc, n, c, st));
	if (nc > nrm_normvar_device_covariates()) {
		// 9 .. 32 covariates (round 6; the entry answered NRM_E_UNSUPPORTED): the Gram-launch form of normalisr_amd/norm.py without torch -- U = e^2 and V = e^2 y
		// (nrm_normvar_weights), M_g = (U P^T)_g with P the nc (nc + 1) / 2 products of covariate rows and a_g = (V C^T)_g on the fp64 Gram kernel, the genes'
		// pseudo-inverses by the library's threaded Jacobi stack (inv_rank's rule: norm.py:152-160), b_g = M_g^+ a_g, the variance-keeping scale (norm.py:248-259),
		// one pass writes the result.
		const int64_t npair = nc * (nc + 1) / 2, rp = nrm_round_up(rows, NRM_ROW_TILE), kp = nrm_round_up(n, NRM_K_TILE), pp = nrm_round_up(npair, NRM_ROW_TILE),
					  cp = nrm_round_up(nc, NRM_ROW_TILE);
		DevBuf u, v, s1, s2, pr, cpad, gm, ga, gwork;
		NRM_TRY(u.alloc((size_t)rp * kp * 8));
		NRM_TRY(v.alloc((size_t)rp * kp * 8));
		NRM_TRY(s1.alloc((size_t)rp * 8));
		NRM_TRY(s2.alloc((size_t)rp * 8));
		NRM_TRY(nrm_normvar_weights(y.p, y_dtype, rows, n, n, lnw.as<double>(), wt.as<double>(), u.as<double>(), v.as<double>(), kp, rp, s1.as<double>(), s2.as<double>(), st));
		{  // the operands of the two contractions, built on the host from the covariates (a few MB)
			const double* hc = h_c;
			std::vector<double> hp((size_t)pp * kp, 0.0), hcp((size_t)cp * kp, 0.0);
			int64_t j = 0;
			for (int64_t a = 0; a < nc; a++)
				for (int64_t d = a; d < nc; d++, j++)
					for (int64_t k = 0; k < n; k++) hp[(size_t)(j * kp + k)] = hc[a * n + k] * hc[d * n + k];
			for (int64_t a = 0; a < nc; a++) memcpy(&hcp[(size_t)(a * kp)], hc + a * n, (size_t)n * 8);
			NRM_TRY(pr.alloc(hp.size() * 8));
			NRM_TRY(cpad.alloc(hcp.size() * 8));
			NRM_HIP(hipMemcpy(pr.p, hp.data(), hp.size() * 8, hipMemcpyHostToDevice));
			NRM_HIP(hipMemcpy(cpad.p, hcp.data(), hcp.size() * 8, hipMemcpyHostToDevice));
		}
		NRM_TRY(gm.alloc((size_t)rp * pp * 8));
		NRM_TRY(ga.alloc((size_t)rp * cp * 8));
		NRM_TRY(gwork.alloc((size_t)nrm_gram_workspace_bytes()));
		NRM_TRY(nrm_gram_f64(u.as<double>(), pr.as<double>(), rp, pp, kp, kp, kp, gm.as<double>(), pp, 0, rows, npair, gwork.p, st));
		NRM_TRY(nrm_gram_f64(v.as<double>(), cpad.as<double>(), rp, cp, kp, kp, kp, ga.as<double>(), cp, 0, rows, nc, gwork.p, st));
		std::vector<double> hgm, hga, hs1, hs2;
		NRM_TRY(download(hgm, gm.p, (size_t)rp * pp));
		NRM_TRY(download(hga, ga.p, (size_t)rp * cp));
		NRM_TRY(download(hs1, s1.p, (size_t)rows));
		NRM_TRY(download(hs2, s2.p, (size_t)rows));
		u.release();
		v.release();
		std::vector<double> m((size_t)rows * nc * nc), mi((size_t)rows * nc * nc), hb((size_t)rows * nc), hscale((size_t)rows, 1.0);
		std::vector<int64_t> rk((size_t)rows, 0);
		for (int64_t g = 0; g < rows; g++) {
			int64_t j = 0;
			for (int64_t a = 0; a < nc; a++)
				for (int64_t d = a; d < nc; d++, j++) m[(size_t)((g * nc + a) * nc + d)] = m[(size_t)((g * nc + d) * nc + a)] = hgm[(size_t)(g * pp + j)];
		}
		NRM_TRY(nrm_small_pinv(m.data(), rows, nc, tol, mi.data(), rk.data(), 0));
		int64_t zr = 0;
		for (int64_t g = 0; g < rows; g++) zr += rk[(size_t)g] <= 0;
		*zero_rank = zr;
		if (zr) return NRM_OK;
		for (int64_t g = 0; g < rows; g++) {
			const double* a = &hga[(size_t)(g * cp)];
			double ab = 0.0;
			for (int64_t q = 0; q < nc; q++) {
				double t = 0.0;
				for (int64_t d = 0; d < nc; d++) t += mi[(size_t)((g * nc + q) * nc + d)] * a[d];
				hb[(size_t)(g * nc + q)] = t;
				ab += a[q] * t;
			}
			if (keepvar) {
				const double mean = hs1[(size_t)g] / (double)n;
				const double dv = std::sqrt(std::fmax(hs2[(size_t)g] / (double)n - mean * mean, 0.0));  // norm.py:248-249
				const double dv2 = std::sqrt(std::fmax(hs2[(size_t)g] - ab, 0.0) / (double)n);           // |y' - P y'|^2 = |y'|^2 - a . b
				hscale[(size_t)g] = std::pow(dv / dv2, h_wt[g]);                                        // norm.py:259
			}
		}
		NRM_TRY(b.alloc((size_t)rows * nc * 8));
		NRM_TRY(scale.alloc((size_t)rows * 8));
		NRM_HIP(hipMemcpy(b.p, hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
		NRM_HIP(hipMemcpy(scale.p, hscale.data(), hscale.size() * 8, hipMemcpyHostToDevice));
		NRM_TRY(flags.alloc(16));
		NRM_HIP(hipMemsetAsync(flags.p, 0, 16, st));
		const size_t ob = (size_t)rows * n * nrm_esize(out_dtype);
		NRM_TRY(out.alloc(ob));
		NRM_TRY(nrm_normvar_apply(y.p, y_dtype, rows, n, n, lnw.as<double>(), wt.as<double>(), c.as<double>(), nc, n, b.as<double>(), scale.as<double>(), out.p, out_dtype, n,
								  flags.as<int32_t>(), st));
		int32_t hf[4];
		NRM_HIP(hipMemcpyAsync(hf, flags.p, 16, hipMemcpyDeviceToHost, st));
		NRM_HIP(hipStreamSynchronize(st));
		if (hf[1]) {
			nrm_set_error("normvar: non-finite results (norm.py:286)");
			return NRM_E_NUMERIC;
		}
		return copy_out(h_out, out.p, ob);
	}
	NRM_TRY(mom.alloc((size_t)rows * (size_t)(nc * (nc + 1) / 2 + nc + 2) * 8));
	NRM_TRY(b.alloc((size_t)rows * nc * 8));
	NRM_TRY(scale.alloc((size_t)rows * 8));
	NRM_TRY(rank.alloc((size_t)rows * 8));
	NRM_TRY(flags.alloc(16));
	NRM_HIP(hipMemsetAsync(flags.p, 0, 16, st));
	NRM_TRY(nrm_normvar_solve(y.p, y_dtype, rows, n, n, lnw.as<double>(), wt.as<double>(), c.as<double>(), nc, n, tol, keepvar ? 1 : 0, mom.as<double>(), b.as<double>(),
							  scale.as<double>(), rank.as<int64_t>(), flags.as<int32_t>(), st));
	const size_t ob = (size_t)rows * n * nrm_esize(out_dtype);
	NRM_TRY(out.alloc(ob));
	NRM_TRY(nrm_normvar_apply(y.p, y_dtype, rows, n, n, lnw.as<double>(), wt.as<double>(), c.as<double>(), nc, n, b.as<double>(), scale.as<double>(), out.p, out_dtype, n,
							  flags.as<int32_t>(), st));
	int32_t hf[4];
	NRM_HIP(hipMemcpyAsync(hf, flags.p, 16, hipMemcpyDeviceToHost, st));
	NRM_HIP(hipStreamSynchronize(st));
	*zero_rank = hf[0];
	if (hf[0]) return NRM_OK;
	if (hf[1]) {
		nrm_set_error("normvar: non-finite results (norm.py:286)");
		return NRM_E_NUMERIC;
	}
	return copy_out(h_out, out.p, ob);
}

// ---- single=4 (association.py:421-576,926-980) in closed form, for full-rank designs --------------------------------------------------------
namespace {

// Inverse of the symmetric positive definite matrix whose upper tiles are in d_m (nxp x nxp) by Newton-Schulz iteration on the fp64 Gram kernel
// (normalisr_amd/single4.py: _spd_inverse_device); d_n receives the inverse (zero padding), small (3, nx) its diagonal / kappa numerators / absolute
// row sums; *norm1 = ||M||_1.  *ok = 0: not converged (the caller takes the host's Cholesky factorisation).
int spd_inverse_device(const double* d_m, int64_t nx, int64_t nxp, const double* d_ss, DevBuf& d_n, std::vector<double>& small, double* norm1, int* ok, void* gwork,
					   hipStream_t st) {
	*ok = 0;
	DevBuf mp, t, tt, xt, x, scal, work, res, dsmall;
	const size_t mb = (size_t)nxp * nxp * 8;
	NRM_TRY(mp.alloc(mb));
	NRM_TRY(t.alloc(mb));
	NRM_TRY(tt.alloc(mb));
	NRM_TRY(xt.alloc(mb));
	NRM_TRY(x.alloc(mb));
	NRM_TRY(scal.alloc(16));
	NRM_TRY(work.alloc((size_t)std::max<int64_t>(2 * nxp, (nxp / 32) * (nxp / 32)) * 8));
	NRM_TRY(res.alloc(8));
	NRM_TRY(nrm_spd_prepare(d_m, nxp, nx, nxp, mp.as<double>(), scal.as<double>(), work.as<double>(), st));
	double hs[2];
	NRM_HIP(hipMemcpyAsync(hs, scal.p, 16, hipMemcpyDeviceToHost, st));
	NRM_HIP(hipStreamSynchronize(st));
	*norm1 = hs[0];
	if (!std::isfinite(hs[0]) || hs[0] <= 0) return NRM_OK;
	bool done = false;
	for (int start = 0; start < 2 && !done; start++) {  // diag(1 / M_ii) first (nearly orthogonal rows: five steps), then I / ||M||_1 (always converges)
		NRM_TRY(nrm_spd_start(mp.as<double>(), nxp, start == 0 ? 1 : 0, scal.as<double>(), x.as<double>(), st));
		const int look_from = start == 0 ? 2 : 4;
		bool diverged = false;
		for (int it = 0; it < 60; it++) {
			NRM_TRY(nrm_gram_f64(mp.as<double>(), x.as<double>(), nxp, nxp, nxp, nxp, nxp, t.as<double>(), nxp, 0, 0, 0, gwork, st));  // T = M X
			NRM_TRY(nrm_spd_transpose_residual(t.as<double>(), nxp, tt.as<double>(), res.as<double>(), work.as<double>(), st));
			double r = INFINITY;
			if (it >= look_from) {
				NRM_HIP(hipMemcpyAsync(&r, res.p, 8, hipMemcpyDeviceToHost, st));
				NRM_HIP(hipStreamSynchronize(st));
			}
			if (std::isnan(r) || (start == 0 && it == look_from && !(r < 1.0))) {
				diverged = true;
				break;
			}
			NRM_TRY(nrm_gram_f64(x.as<double>(), tt.as<double>(), nxp, nxp, nxp, nxp, nxp, xt.as<double>(), nxp, 0, 0, 0, gwork, st));  // X T
			NRM_TRY(nrm_spd_update(x.as<double>(), xt.as<double>(), nxp * nxp, st));
			if (r < 1e-7) {
				done = true;
				break;
			}
		}
		if (start == 1 && diverged) return NRM_OK;
	}
	if (!done) return NRM_OK;
	NRM_TRY(d_n.alloc(mb));
	NRM_TRY(dsmall.alloc((size_t)3 * nx * 8));
	NRM_TRY(nrm_spd_finish(x.as<double>(), nx, nxp, d_ss, d_n.as<double>(), dsmall.as<double>(), st));
	NRM_TRY(download(small, dsmall.p, (size_t)3 * nx));
	*ok = 1;
	return NRM_OK;
}

// the same on the host (fallback): Cholesky factor, triangular inverse, L^-T L^-1; false when the matrix is not positive definite
bool spd_inverse_host(std::vector<double>& m, int64_t n) {
	std::vector<double> l((size_t)n * n, 0.0);
	for (int64_t j = 0; j < n; j++) {
		double d = m[(size_t)(j * n + j)];
		for (int64_t k = 0; k < j; k++) d -= l[(size_t)(j * n + k)] * l[(size_t)(j * n + k)];
		if (!(d > 0)) return false;
		l[(size_t)(j * n + j)] = std::sqrt(d);
		for (int64_t i = j + 1; i < n; i++) {
			double s = m[(size_t)(i * n + j)];
			for (int64_t k = 0; k < j; k++) s -= l[(size_t)(i * n + k)] * l[(size_t)(j * n + k)];
			l[(size_t)(i * n + j)] = s / l[(size_t)(j * n + j)];
		}
	}
	std::vector<double> li((size_t)n * n, 0.0);  // L^-1, lower triangular
	for (int64_t j = 0; j < n; j++) {
		li[(size_t)(j * n + j)] = 1.0 / l[(size_t)(j * n + j)];
		for (int64_t i = j + 1; i < n; i++) {
			double s = 0.0;
			for (int64_t k = j; k < i; k++) s -= l[(size_t)(i * n + k)] * li[(size_t)(k * n + j)];
			li[(size_t)(i * n + j)] = s / l[(size_t)(i * n + i)];
		}
	}
	for (int64_t i = 0; i < n; i++)
		for (int64_t j = 0; j <= i; j++) {
			double s = 0.0;
			for (int64_t k = i; k < n; k++) s += li[(size_t)(k * n + i)] * li[(size_t)(k * n + j)];
			m[(size_t)(i * n + j)] = m[(size_t)(j * n + i)] = s;
		}
	return true;
}

}  // namespace

extern "C" int nrm_association_tests_single4_host(const void* h_dx, int x_dtype, int64_t nx, const void* h_dy, int y_dtype, int64_t ny, const void* h_dc, int c_dtype,
												   int64_t nc, int64_t n, const double* h_dci, int rank, int dimreduce, int return_dot, double tol, void* h_p,
												   void* h_stat, void* h_alpha, void* h_varx, void* h_vary, int out_dtype) {
	std::lock_guard<std::mutex> serial(nrm_host_entry_mutex());
	NRM_TRY(nrm_bind_device());
	NRM_REQUIRE(h_dx && h_dy && nx > 0 && ny > 0 && n > 0 && nc >= 0 && (nc == 0 || (h_dc && h_dci)), "Unmatching dx/dy/dc dimensions.");
	NRM_REQUIRE(h_p && h_stat && h_varx && h_vary, "nrm_association_tests_single4_host: null output");
	NRM_REQUIRE((x_dtype == NRM_F32 || x_dtype == NRM_F64) && (y_dtype == NRM_F32 || y_dtype == NRM_F64) && (out_dtype == NRM_F32 || out_dtype == NRM_F64), "bad dtype");
	const int64_t m = nx + nc;
	if (rank != nc) {  // (a rank-deficient C C^T is a principal block of A A^T: no closed form)
		nrm_set_error("nrm_association_tests_single4_host covers full-rank designs (closed form); rank-deficient covariates follow the per-grouping algorithm of the package");
		return NRM_E_UNSUPPORTED;
	}
	if (n <= m + dimreduce) {
		nrm_set_error("Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.");
		return NRM_E_DEVICE;
	}
	hipStream_t st = nullptr;
	const int64_t kp = nrm_round_up(n, NRM_K_TILE), nxp = nrm_round_up(nx, NRM_ROW_TILE), nyp = nrm_round_up(ny, NRM_ROW_TILE);
	DevBuf dx, dy, dc, dci, gwork, flags;
	std::vector<double> c64;
	NRM_TRY(upload_matrix(h_dx, x_dtype, nx, n, dx, st));
	NRM_TRY(upload_matrix(h_dy, y_dtype, ny, n, dy, st));
	NRM_TRY(covariates_f64(h_dc, c_dtype, nc, n, c64, dc));
	if (nc) {
		NRM_TRY(dci.alloc((size_t)nc * nc * 8));
		NRM_HIP(hipMemcpy(dci.p, h_dci, (size_t)nc * nc * 8, hipMemcpyHostToDevice));
	}
	NRM_TRY(gwork.alloc((size_t)nrm_gram_workspace_bytes()));
	NRM_TRY(flags.alloc(16));
	NRM_HIP(hipMemsetAsync(flags.p, 0, 16, st));
	double cval = 0.0;
	const int ci = nc ? constant_row(c64.data(), nc, n, &cval) : -1;
	// the design rows: from their entries when there are few (gRNA incidence), else K1's fp64 residuals
	DevBuf ssx, bx, mt, rxd;
	NRM_TRY(ssx.alloc((size_t)nxp * 8));
	NRM_HIP(hipMemsetAsync(ssx.p, 0, (size_t)nxp * 8, st));
	NRM_TRY(bx.alloc((size_t)nx * (nc > 0 ? nc : 1) * 8));
	NRM_HIP(hipMemsetAsync(bx.p, 0, (size_t)nx * (nc > 0 ? nc : 1) * 8, st));
	NRM_TRY(mt.alloc((size_t)nxp * nxp * 8));
	NrmDesignLists L;
	bool sparse = false;
	{
		const char* mode = getenv("NRM_DE_SPARSE");
		const bool off = mode && !strcmp(mode, "0"), force = mode && !strcmp(mode, "force");
		if (!off && nc <= nrm_de_sparse_max_covariates() && (force || (nx >= 32 && ny >= 64 && n >= 2048 && nx * n >= (1ll << 22)))) {
			NRM_TRY(L.build(dx.p, x_dtype, nx, n, true, 1.0 / 16, st));
			sparse = L.ok;
		}
	}
	for (int pass = 0; pass < 2; pass++) {  // (a second pass only when the sparse-design kernels hand rows back: the same on K1 and the fp64 Gram kernel)
		if (sparse) {
			NRM_TRY(nrm_design_stats(L.row_ptr.as<int64_t>(), L.cells.as<int32_t>(), L.row_vals.as<double>(), dc.as<double>(), n, nc, dci.as<double>(), nx, ssx.as<double>(),
									 nc ? bx.as<double>() : nullptr, flags.as<int32_t>(), st));
			DevBuf ss2;  // (|x~|^2 again, as the product kernel computes it for its rows: not used)
			NRM_TRY(ss2.alloc((size_t)nxp * 8));
			// M~ = X~ X~^T: the same product with the design rows in the place of the expression rows
			NRM_TRY(sparse_products(L, dx.p, x_dtype, nx, n, dc.as<double>(), nc, ci, cval, dci.as<double>(), bx.as<double>(), nc > 0 ? nc : 1, mt.as<double>(), nxp, 0,
									ss2.as<double>(), nullptr, flags.as<int32_t>(), st));
			// a design row all but inside the span of the covariates is known HERE: looked at before the inverse and the genes' products, which a
			// handed-back call would only repeat (round-5 advisory)
			int32_t hf[4];
			NRM_HIP(hipMemcpyAsync(hf, flags.p, 16, hipMemcpyDeviceToHost, st));
			NRM_HIP(hipStreamSynchronize(st));
			if (hf[2] > 0) {
				sparse = false;
				NRM_HIP(hipMemsetAsync(flags.p, 0, 16, st));
				continue;
			}
		} else {
			NRM_TRY(rxd.alloc((size_t)nxp * kp * 8));
			NRM_TRY(nrm_residualize(dx.p, x_dtype, nx, n, n, dc.as<double>(), nc, n, dci.as<double>(), rank, rxd.as<double>(), kp, nxp, ssx.as<double>(),
									nc ? bx.as<double>() : nullptr, st));
			NRM_TRY(nrm_gram_f64(rxd.as<double>(), rxd.as<double>(), nxp, nxp, kp, kp, kp, mt.as<double>(), nxp, 1, nx, nx, gwork.p, st));
		}
		// N~ = M~^-1
		DevBuf dn;
		std::vector<double> small;
		double norm_mt = 0.0;
		int okdev = 0;
		NRM_TRY(spd_inverse_device(mt.as<double>(), nx, nxp, ssx.as<double>(), dn, small, &norm_mt, &okdev, gwork.p, st));
		if (!okdev) {
			std::vector<double> hm, hss;
			NRM_TRY(download(hm, mt.p, (size_t)nxp * nxp));
			NRM_TRY(download(hss, ssx.p, (size_t)nx));
			std::vector<double> a((size_t)nx * nx);
			for (int64_t i = 0; i < nx; i++)
				for (int64_t j = 0; j < nx; j++) a[(size_t)(i * nx + j)] = i <= j ? hm[(size_t)(i * nxp + j)] : hm[(size_t)(j * nxp + i)];
			norm_mt = 0.0;
			for (int64_t j = 0; j < nx; j++) {
				double s = 0.0;
				for (int64_t i = 0; i < nx; i++) s += std::fabs(a[(size_t)(i * nx + j)]);
				norm_mt = s > norm_mt ? s : norm_mt;
			}
			if (!spd_inverse_host(a, nx)) {
				nrm_set_error("design rows are linearly dependent given the covariates");
				return NRM_E_UNSUPPORTED;
			}
			std::vector<double> npad((size_t)nxp * nxp, 0.0);
			small.assign((size_t)3 * nx, 0.0);
			for (int64_t i = 0; i < nx; i++) {
				for (int64_t j = 0; j < nx; j++) {
					const double v = a[(size_t)(i * nx + j)];
					npad[(size_t)(i * nxp + j)] = v;
					small[(size_t)(nx + i)] += std::fabs(v) * std::sqrt(hss[(size_t)j]);
					small[(size_t)(2 * nx + i)] += std::fabs(v);
				}
				small[(size_t)i] = a[(size_t)(i * nx + i)];
			}
			NRM_TRY(dn.alloc(npad.size() * 8));
			NRM_HIP(hipMemcpy(dn.p, npad.data(), npad.size() * 8, hipMemcpyHostToDevice));
		}
		double norm_ninv = 0.0;
		std::vector<double> dxx((size_t)nx);
		for (int64_t i = 0; i < nx; i++) {
			const double d = small[(size_t)i];
			if (!std::isfinite(d) || !std::isfinite(small[(size_t)(nx + i)]) || !std::isfinite(small[(size_t)(2 * nx + i)]) || !(d > 0)) {
				nrm_set_error("design rows are linearly dependent given the covariates");
				return NRM_E_UNSUPPORTED;
			}
			norm_ninv = small[(size_t)(2 * nx + i)] > norm_ninv ? small[(size_t)(2 * nx + i)] : norm_ninv;
			dxx[(size_t)i] = 1.0 / ((double)n * d);
		}
		// the genes: Y~ X~^T (as (genes, design rows)), |y~|^2, b_y on request
		const bool want_alpha = h_alpha != nullptr && nc > 0;
		DevBuf g, ssy, by, ryd;
		NRM_TRY(g.alloc((size_t)nyp * nxp * 8));
		NRM_HIP(hipMemsetAsync(g.p, 0, (size_t)nyp * nxp * 8, st));
		NRM_TRY(ssy.alloc((size_t)nyp * 8));
		if (want_alpha) {
			NRM_TRY(by.alloc((size_t)ny * nc * 8));
			NRM_HIP(hipMemsetAsync(by.p, 0, (size_t)ny * nc * 8, st));
		}
		if (sparse) {
			NRM_TRY(sparse_products(L, dy.p, y_dtype, ny, n, dc.as<double>(), nc, ci, cval, dci.as<double>(), bx.as<double>(), nc > 0 ? nc : 1, g.as<double>(), nxp, 1,
									ssy.as<double>(), want_alpha ? by.as<double>() : nullptr, flags.as<int32_t>(), st));
			int32_t hf[4];
			NRM_HIP(hipMemcpy(hf, flags.p, 16, hipMemcpyDeviceToHost));
			if (hf[2] > 0) {  // rows all but inside the span of the covariates: K1's two sweeps and the fp64 Gram kernel for this call
				sparse = false;
				NRM_HIP(hipMemsetAsync(flags.p, 0, 16, st));
				continue;
			}
		} else {
			if (!rxd.p) {  // (handed back from the sparse kernels: the design rows' residuals are needed after all)
				NRM_TRY(rxd.alloc((size_t)nxp * kp * 8));
				NRM_TRY(nrm_residualize(dx.p, x_dtype, nx, n, n, dc.as<double>(), nc, n, dci.as<double>(), rank, rxd.as<double>(), kp, nxp, ssx.as<double>(),
										nc ? bx.as<double>() : nullptr, st));
			}
			NRM_TRY(ryd.alloc((size_t)nyp * kp * 8));
			NRM_TRY(nrm_residualize(dy.p, y_dtype, ny, n, n, dc.as<double>(), nc, n, dci.as<double>(), rank, ryd.as<double>(), kp, nyp, ssy.as<double>(),
									want_alpha ? by.as<double>() : nullptr, st));
			NRM_TRY(nrm_gram_f64(ryd.as<double>(), rxd.as<double>(), nyp, nxp, kp, kp, kp, g.as<double>(), nxp, 0, ny, nx, gwork.p, st));
			ryd.release();
		}
		// B^T = (Y~ X~^T) N~, the sweep
		DevBuf bt, ddxx, op, ostat, ovary, work;
		NRM_TRY(bt.alloc((size_t)nyp * nxp * 8));
		NRM_TRY(nrm_gram_f64(g.as<double>(), dn.as<double>(), nyp, nxp, nxp, nxp, nxp, bt.as<double>(), nxp, 0, ny, nx, gwork.p, st));
		NRM_TRY(ddxx.alloc((size_t)nx * 8));
		NRM_HIP(hipMemcpyAsync(ddxx.p, dxx.data(), (size_t)nx * 8, hipMemcpyHostToDevice, st));
		const size_t ob = (size_t)nx * ny * nrm_esize(out_dtype);
		NRM_TRY(op.alloc(ob));
		NRM_TRY(ostat.alloc(ob));
		NRM_TRY(ovary.alloc(ob));
		NRM_TRY(work.alloc((size_t)ny * 8));
		NRM_TRY(nrm_single4_sweep(bt.as<double>(), g.as<double>(), nxp, ssy.as<double>(), ddxx.as<double>(), nx, ny, nx, n, (double)(n - m - dimreduce), return_dot, op.p, ostat.p,
								  ovary.p, out_dtype, ny, work.as<double>(), flags.as<int32_t>(), st));
		// (the reference's assertions on the closed form's results speak only if the closed form applies: a nearly rank-deficient design can fail them
		//  where the per-grouping algorithm -- which the package then takes -- returns results; association.py:421-576.  Round-5 advisory.)
		const int flag_rc = check_flags2(flags.as<int32_t>(), st);
		if (flag_rc == NRM_E_DEVICE) return flag_rc;
		// Does the closed form apply?  The reference's own rank test on A A^T (singular values >= tol x the largest, association.py:77), settled from
		// norms at hand (single4.py: _surely_full_rank): lambda_max <= ||M~||_1 + ||a||_F^2 ||Mcc^-1|| + ||Mcc||, 1 / lambda_min <= ||N~||_1 (1 + ||b||_F)^2 + ||Mcc^-1||
		double lam_max = norm_mt, inv_norm = norm_ninv;
		std::vector<double> hbx;
		if (nc) {
			std::vector<double> mcc((size_t)nc * nc, 0.0), ev((size_t)nc);
			for (int64_t c = 0; c < nc; c++)
				for (int64_t d = c; d < nc; d++) {
					double s = 0.0;
					for (int64_t k = 0; k < n; k++) s += c64[(size_t)(c * n + k)] * c64[(size_t)(d * n + k)];
					mcc[(size_t)(c * nc + d)] = mcc[(size_t)(d * nc + c)] = s;
				}
			NRM_TRY(nrm_small_eigvals(mcc.data(), nc, ev.data()));
			NRM_TRY(download(hbx, bx.p, (size_t)nx * nc));
			double a2 = 0.0, b2 = 0.0;
			for (int64_t i = 0; i < nx; i++)
				for (int64_t c = 0; c < nc; c++) {
					double a = 0.0;
					for (int64_t d = 0; d < nc; d++) a += hbx[(size_t)(i * nc + d)] * mcc[(size_t)(d * nc + c)];
					a2 += a * a;
					b2 += hbx[(size_t)(i * nc + c)] * hbx[(size_t)(i * nc + c)];
				}
			if (!(ev[0] > 0)) lam_max = NAN;
			else {
				lam_max = norm_mt + a2 / ev[0] + ev[(size_t)nc - 1];
				inv_norm = norm_ninv * (1.0 + std::sqrt(b2)) * (1.0 + std::sqrt(b2)) + 1.0 / ev[0];
			}
		}
		if (!(std::isfinite(lam_max) && std::isfinite(inv_norm) && lam_max > 0 && inv_norm > 0 && 1.0 / (inv_norm * lam_max) >= 2.0 * tol)) {
			nrm_set_error("nrm_association_tests_single4_host: the design may be rank deficient at tol = %g (no certificate from the norms); the package takes the spectrum of A A^T and, if need be, the per-grouping algorithm", tol);
			return NRM_E_UNSUPPORTED;
		}
		if (flag_rc) {
			nrm_set_error("association results failed the reference's assertions (association.py:557): non-finite values or R^2 > 1+1e-8 in the closed form of a full-rank design");
			return flag_rc;
		}
		NRM_TRY(copy_out(h_p, op.p, ob));
		NRM_TRY(copy_out(h_stat, ostat.p, ob));
		NRM_TRY(copy_out(h_vary, ovary.p, ob));
		for (int64_t i = 0; i < nx; i++) {
			const double v = dxx[(size_t)i] == 0.0 ? 1.0 : dxx[(size_t)i];
			if (out_dtype == NRM_F64)
				((double*)h_varx)[i] = v;
			else
				((float*)h_varx)[i] = (float)v;
		}
		if (want_alpha) {
			// alpha_y = b_y - B_y b_x, the same for every grouping (association.py:551-553 in the closed form): B^T (ny, nx) against b_x (nx, nc) on the host
			std::vector<double> hbt, hby;
			NRM_TRY(download(hbt, bt.p, (size_t)nyp * nxp));
			NRM_TRY(download(hby, by.p, (size_t)ny * nc));
			const size_t esz = nrm_esize(out_dtype);
			for (int64_t y = 0; y < ny; y++)
				for (int64_t c = 0; c < nc; c++) {
					double s = hby[(size_t)(y * nc + c)];
					for (int64_t i = 0; i < nx; i++) s -= hbt[(size_t)(y * nxp + i)] * hbx[(size_t)(i * nc + c)];
					for (int64_t i = 0; i < nx; i++) {
						char* o = (char*)h_alpha + ((size_t)(i * ny + y) * nc + c) * esz;
						if (out_dtype == NRM_F64)
							*(double*)o = s;
						else
							*(float*)o = (float)s;
					}
				}
		}
		return NRM_OK;
	}
	nrm_set_error("nrm_association_tests_single4_host: internal error (no pass completed)");
	return NRM_E_DEVICE;
}
