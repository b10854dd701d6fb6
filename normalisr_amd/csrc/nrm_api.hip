// C ABI glue: error reporting, device selection and the whole-problem host entry
// (numpy buffers in, numpy buffers out) that stands where association_tests() does
// (association.py:761-771,1093) for single=0.
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <strings.h>
#include <vector>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>
#include "nrm_common.h"
#include "nrm_host_logic.h"
#include "nrm_host_entry.h"

static thread_local char g_err[512] = "";

void nrm_set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

extern "C" int nrm_version(void) { return 100; }
extern "C" const char* nrm_last_error(void) { return g_err; }

extern "C" int nrm_device_count(int* count) {
	NRM_REQUIRE(count != nullptr, "nrm_device_count: null pointer");
	NRM_HIP(hipGetDeviceCount(count));
	return NRM_OK;
}

// The GPU the library's whole-problem entries run on.  hipSetDevice is per THREAD: the entry a Python thread calls, the helper threads that page-lock
// the caller's result arrays beside the kernels, and a second Python thread calling the next entry must all bind to the same device, and the scratch pool
// and the upload ring hold memory OF a device.  nrm_set_device records the choice for the process (validated: NRM_E_ARG -> ValueError for an index
// that is not there), releases what was cached for another device, and binds the calling thread; nrm_bind_device() binds any other thread.
static std::atomic<int> g_device{-1};
static NrmDevPool g_pool;

extern "C" int nrm_set_device(int device) {
	int count = 0;
	NRM_HIP(hipGetDeviceCount(&count));
	NRM_REQUIRE(device >= 0 && device < count, "GPU %d requested, %d visible", device, count);
	const int prev = g_device.exchange(device);
	if (prev != device && prev >= 0) {  // (scratch blocks and page-locked ring slots belong to the device they were made on)
		(void)hipSetDevice(prev);
		(void)hipDeviceSynchronize();
		g_pool.release();
		(void)nrm_upload_release();
	}
	NRM_HIP(hipSetDevice(device));
	return NRM_OK;
}

int nrm_bind_device(void) {
	const int d = g_device.load();
	if (d >= 0) NRM_HIP(hipSetDevice(d));
	return NRM_OK;
}

// h[0:a, a:b] = h[a:b, 0:a]^T on the host, for a row-major matrix of elem_bytes (4 or 8) elements with ld_bytes between rows: the rows
// a..b of a symmetric result matrix have arrived over PCIe up to column b, their mirror image above the diagonal is made here instead
// of travelling too (the reference's gather loop mirrors on the host as well: association.py:1049-1057).  32 x 32 tiles, the destination
// rows shared out over `threads` host threads (0: one per 4 hardware threads, at most 32).
template <typename E>
static void mirror_rows(char* h, int64_t ld, int64_t a, int64_t b, int64_t r0, int64_t r1) {
	constexpr int64_t TB = 32;
	for (int64_t i0 = r0; i0 < r1; i0 += TB)      // destination rows (columns of the source block)
		for (int64_t j0 = a; j0 < b; j0 += TB) {   // destination columns (rows of the source block)
			const int64_t i1 = std::min(i0 + TB, r1), j1 = std::min(j0 + TB, b);
			for (int64_t i = i0; i < i1; i++) {
				E* dst = reinterpret_cast<E*>(h + i * ld) + j0;
				for (int64_t j = j0; j < j1; j++) dst[j - j0] = reinterpret_cast<const E*>(h + j * ld)[i];
			}
		}
}

extern "C" int nrm_host_mirror_rows(void* h, int64_t ld_bytes, int elem_bytes, int64_t a, int64_t b, int threads) {
	NRM_REQUIRE(h && (elem_bytes == 4 || elem_bytes == 8) && 0 <= a && a <= b && ld_bytes >= b * elem_bytes, "nrm_host_mirror_rows: bad arguments");
	if (a == 0 || a == b) return NRM_OK;
	if (threads <= 0) threads = (int)std::min<int64_t>(32, std::max<int64_t>(1, std::thread::hardware_concurrency() / 4));
	threads = (int)std::min<int64_t>(threads, (a + 31) / 32);
	auto work = [&](int64_t r0, int64_t r1) {
		if (elem_bytes == 4)
			mirror_rows<uint32_t>((char*)h, ld_bytes, a, b, r0, r1);
		else
			mirror_rows<uint64_t>((char*)h, ld_bytes, a, b, r0, r1);
	};
	if (threads <= 1) {
		work(0, a);
		return NRM_OK;
	}
	const int64_t per = ((a + threads - 1) / threads + 31) / 32 * 32;
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; t++) {
		const int64_t r0 = t * per, r1 = std::min<int64_t>(a, r0 + per);
		if (r1 > r0) pool.emplace_back(work, r0, r1);
	}
	for (auto& th : pool) th.join();
	return NRM_OK;
}

extern "C" int nrm_host_pin(void* ptr, int64_t bytes, int threads) {
	NRM_REQUIRE(ptr != nullptr && bytes > 0, "nrm_host_pin: empty range");
	NRM_TRY_RC(nrm_bind_device());  // (called from the entries' helper threads too: a fresh thread's current device is 0)
	const int64_t page = 4096;
	if (threads <= 0) {
		threads = (int)std::min<int64_t>(16, std::max<int64_t>(1, bytes / (8 << 20)));
		threads = (int)std::min<int64_t>(threads, std::max(1u, std::thread::hardware_concurrency()));
	}
	// first touch from several threads: page faults of a fresh mapping are the cost of page-locking it (a single thread
	// faults ~13 GB/s, hipHostRegister of an untouched range ~22 GB/s, 16 threads > 100 GB/s)
	auto touch = [=](int64_t lo, int64_t hi) {
		volatile char* q = (volatile char*)ptr;
		for (int64_t o = lo; o < hi; o += page) q[o] = q[o];
		if (hi > lo) q[hi - 1] = q[hi - 1];
	};
	if (threads == 1) {
		touch(0, bytes);
	} else {
		const int64_t chunk = ((bytes + threads - 1) / threads + page - 1) / page * page;
		std::vector<std::thread> pool;
		for (int t = 0; t < threads; t++) {
			const int64_t lo = t * chunk, hi = std::min<int64_t>(bytes, lo + chunk);
			if (hi > lo) pool.emplace_back(touch, lo, hi);
		}
		for (auto& th : pool) th.join();
	}
	NRM_HIP(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
	return NRM_OK;
}

extern "C" int nrm_host_unpin(void* ptr) {
	NRM_REQUIRE(ptr != nullptr, "nrm_host_unpin: null pointer");
	NRM_HIP(hipHostUnregister(ptr));
	return NRM_OK;
}

extern "C" int nrm_host_alloc(void** ptr, int64_t bytes) {
	NRM_REQUIRE(ptr != nullptr && bytes > 0, "nrm_host_alloc: empty request");
	NRM_HIP(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
	return NRM_OK;
}

extern "C" int nrm_host_free(void* ptr) {
	if (ptr) NRM_HIP(hipHostFree(ptr));
	return NRM_OK;
}

extern "C" int nrm_copy_to_host(void* h_dst, const void* d_src, int64_t bytes, void* stream) {
	NRM_REQUIRE(bytes >= 0, "nrm_copy_to_host: negative size");
	if (bytes == 0) return NRM_OK;
	NRM_REQUIRE(h_dst && d_src, "nrm_copy_to_host: null pointer");
	NRM_HIP(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
	return NRM_OK;
}

// (Measured, tools/time_duplex.py: on this platform a 1-D hipMemcpy runs at 56 GB/s in either direction but an upload and a
// download queue behind each other -- 200 MB + 200 MB take 7.2 ms together; this rectangular copy is served by a blit kernel at
// 27 GB/s that does run beside an upload, 4.5 ms together; a hand-written kernel storing into mapped host memory reaches the same
// 27 GB/s with 16 to 4096 workgroups and was slower end to end, it competes with K2 for the CUs.)
extern "C" int nrm_copy_rect_to_host(void* h_dst, int64_t dst_pitch, const void* d_src, int64_t src_pitch, int64_t row_bytes, int64_t rows,
									 void* stream) {
	NRM_REQUIRE(row_bytes >= 0 && rows >= 0 && dst_pitch >= row_bytes && src_pitch >= row_bytes, "nrm_copy_rect_to_host: pitches smaller than the row");
	if (row_bytes == 0 || rows == 0) return NRM_OK;
	NRM_REQUIRE(h_dst && d_src, "nrm_copy_rect_to_host: null pointer");
	NRM_HIP(hipMemcpy2DAsync(h_dst, (size_t)dst_pitch, d_src, (size_t)src_pitch, (size_t)row_bytes, (size_t)rows, hipMemcpyDeviceToHost,
							 (hipStream_t)stream));
	return NRM_OK;
}

extern "C" int nrm_fill_zero(void* d_dst, int64_t bytes, void* stream) {
	NRM_REQUIRE(bytes >= 0, "nrm_fill_zero: negative size");
	if (bytes == 0) return NRM_OK;
	NRM_REQUIRE(d_dst != nullptr, "nrm_fill_zero: null pointer");
	NRM_HIP(hipMemsetAsync(d_dst, 0, (size_t)bytes, (hipStream_t)stream));
	return NRM_OK;
}

extern "C" int nrm_fill_i32(void* d_dst, int32_t value, int64_t count, void* stream) {
	NRM_REQUIRE(count >= 0, "nrm_fill_i32: negative size");
	if (count == 0) return NRM_OK;
	NRM_REQUIRE(d_dst != nullptr && (uintptr_t)d_dst % 4 == 0, "nrm_fill_i32: bad pointer");
	NRM_HIP(hipMemsetD32Async((hipDeviceptr_t)d_dst, value, (size_t)count, (hipStream_t)stream));
	return NRM_OK;
}

extern "C" int nrm_copy_rows(void* d_dst, int64_t dst_pitch, const void* d_src, int64_t src_pitch, int64_t row_bytes, int64_t rows,
							 void* stream) {
	NRM_REQUIRE(row_bytes >= 0 && rows >= 0 && dst_pitch >= row_bytes && src_pitch >= row_bytes, "nrm_copy_rows: pitches smaller than the row");
	if (row_bytes == 0 || rows == 0) return NRM_OK;
	NRM_REQUIRE(d_dst && d_src, "nrm_copy_rows: null pointer");
	NRM_HIP(hipMemcpy2DAsync(d_dst, (size_t)dst_pitch, d_src, (size_t)src_pitch, (size_t)row_bytes, (size_t)rows, hipMemcpyDeviceToDevice,
							 (hipStream_t)stream));
	return NRM_OK;
}

static std::mutex g_host_entry;
NrmDevPool& nrm_host_pool() { return g_pool; }
std::mutex& nrm_host_entry_mutex() { return g_host_entry; }
namespace {
inline int64_t round_up(int64_t v, int64_t m) { return nrm_round_up(v, m); }
inline size_t esize(int dtype) { return nrm_esize(dtype); }
}  // namespace

namespace {
// Page-lock of a caller-owned result array for the duration of one call; a range that cannot be locked (already
// registered by the caller, locked-memory limit) is simply copied to at the pageable rate.
struct HostPin {
	void* p = nullptr;
	void try_pin(void* q, int64_t bytes) {
		if (q && bytes >= (1 << 20) && nrm_host_pin(q, bytes, 0) == NRM_OK) p = q;
	}
	~HostPin() {
		if (p) {
			(void)hipDeviceSynchronize();
			(void)hipHostUnregister(p);
		}
	}
};
struct CopyStream {
	hipStream_t s = nullptr;
	std::vector<hipEvent_t> events;
	~CopyStream() {
		for (hipEvent_t e : events) (void)hipEventDestroy(e);
		if (s) (void)hipStreamDestroy(s);
	}
};
struct Joiner {
	std::thread& t;
	~Joiner() {
		if (t.joinable()) t.join();
	}
};
}  // namespace

extern "C" int nrm_release_cache(void) {
	g_pool.release();
	return NRM_OK;
}

// Verdict of the integer engine's accuracy guard (csrc/nrm_fix.h) for the last whole-problem call of this thread: pairs it could not
// certify on the first pass (0: the integer engine's results were returned; > 0: the call was redone on the fp64 Gram kernel),
// and the largest error estimate of a P-value (relative) among the pairs it looked at.
static thread_local int64_t g_guard_hits = 0;
static thread_local double g_guard_worst = 0.0;

extern "C" int nrm_last_guard(int64_t* hits, double* worst) {
	if (hits) *hits = g_guard_hits;
	if (worst) *worst = g_guard_worst;
	return NRM_OK;
}

// NRM_DEBUG="key=value,key=value" read the way normalisr_amd/_opts.py reads it: split on ',', key and value trimmed, the key compared without case --
// so that a switch means the same to the package's engine and to the library's own entries.  true: `key` is there with exactly `value`.
static bool nrm_debug_is(const char* key, const char* value) {
	const char* d = getenv("NRM_DEBUG");
	if (!d) return false;
	const size_t kl = strlen(key), vl = strlen(value);
	while (*d) {
		const char* end = strchr(d, ',');
		if (!end) end = d + strlen(d);
		const char* eq = (const char*)memchr(d, '=', (size_t)(end - d));
		if (eq) {
			const char *k0 = d, *k1 = eq, *v0 = eq + 1, *v1 = end;
			while (k0 < k1 && isspace((unsigned char)*k0)) k0++;
			while (k1 > k0 && isspace((unsigned char)k1[-1])) k1--;
			while (v0 < v1 && isspace((unsigned char)*v0)) v0++;
			while (v1 > v0 && isspace((unsigned char)v1[-1])) v1--;
			if ((size_t)(k1 - k0) == kl && (size_t)(v1 - v0) == vl && !strncasecmp(k0, key, kl) && !strncmp(v0, value, vl)) return true;
		}
		d = *end ? end + 1 : end;
	}
	return false;
}

static bool de_path_general() {  // NRM_DEBUG="de_path=general" (or NRM_DE_PATH=general): no streaming de kernel, as in normalisr_amd/_opts.py
	if (nrm_debug_is("de_path", "general")) return true;
	const char* e = getenv("NRM_DE_PATH");
	return e && !strcmp(e, "general");
}

static double guard_tolerance() {
	const char* t = getenv("NRM_I8_GUARD_TOL");  // largest relative change of a P-value the integer engine may cause (0: no guard)
	return t ? atof(t) : 2.5e-7;
}

static int association_tests_host_impl(const void* h_dx, int x_dtype, int64_t nx, const void* h_dy, int y_dtype, int64_t ny,
										  const void* h_dc, int c_dtype, int64_t nc, int64_t n, const double* h_dci, int rank,
										  int dimreduce, int return_dot, void* h_p, void* h_stat, void* h_alpha, void* h_varx,
										  void* h_vary, void* h_r, void* h_t, int out_dtype, int nslices, int64_t* guard_hits, bool allow_sparse);

extern "C" int nrm_association_tests_host(const void* h_dx, int x_dtype, int64_t nx, const void* h_dy, int y_dtype, int64_t ny,
										  const void* h_dc, int c_dtype, int64_t nc, int64_t n, const double* h_dci, int rank,
										  int dimreduce, int return_dot, void* h_p, void* h_stat, void* h_alpha, void* h_varx,
										  void* h_vary, void* h_r, void* h_t, int out_dtype) {
	std::lock_guard<std::mutex> serial(g_host_entry);
	NRM_TRY_RC(nrm_bind_device());
	// K2 engine as in the Python host (NRM_GRAM): exact fixed-point contraction on the int8 matrix cores (6 slices = 46 bits; i8x5: 5
	// slices = 38 bits), or the fp64 matrix-core kernel (f64).  With the integer engine K1 writes the digit planes itself and the
	// fp64 residuals are never stored.
	int nslices = 6;
	if (n < 2048 || n >= (1 << 22)) nslices = 0;  // small problems stay on the fp64 kernel (the integer engine's error in r grows as 1/sqrt(n))
	else if (const char* g = getenv("NRM_GRAM")) {
		if (!strcmp(g, "f64")) nslices = 0;
		else if (!strcmp(g, "i8x5")) nslices = 5;
		else NRM_REQUIRE(!strcmp(g, "i8"), "NRM_GRAM must be i8, i8x5 or f64");
	}
	if (nslices && (n % 4 != 0)) nslices = 0;  // K1's fused quantiser needs 16-byte aligned rows of the (unpadded) host matrices
	g_guard_hits = 0;
	g_guard_worst = 0.0;
	int64_t hits = 0;
	int rc = association_tests_host_impl(h_dx, x_dtype, nx, h_dy, y_dtype, ny, h_dc, c_dtype, nc, n, h_dci, rank, dimreduce, return_dot, h_p, h_stat,
										 h_alpha, h_varx, h_vary, h_r, h_t, out_dtype, nslices, &hits, true);
	if (rc == NRM_OK && hits > 0) {  // pairs the guard could not certify (or rows the sparse-design kernels handed back): the whole call again on the fp64 matrix cores
		g_guard_hits = hits;
		rc = association_tests_host_impl(h_dx, x_dtype, nx, h_dy, y_dtype, ny, h_dc, c_dtype, nc, n, h_dci, rank, dimreduce, return_dot, h_p, h_stat,
										 h_alpha, h_varx, h_vary, h_r, h_t, out_dtype, 0, nullptr, false);
	}
	return rc;
}

static int association_tests_host_impl(const void* h_dx, int x_dtype, int64_t nx, const void* h_dy, int y_dtype, int64_t ny,
										  const void* h_dc, int c_dtype, int64_t nc, int64_t n, const double* h_dci, int rank,
										  int dimreduce, int return_dot, void* h_p, void* h_stat, void* h_alpha, void* h_varx,
										  void* h_vary, void* h_r, void* h_t, int out_dtype, int nslices, int64_t* guard_hits, bool allow_sparse) {
	const bool samexy = (h_dy == nullptr);
	if (samexy) {
		ny = nx;
		y_dtype = x_dtype;
	}
	NRM_REQUIRE(h_dx && nx > 0 && ny > 0 && n > 0, "Incorrect dx/dy/dc size.");
	NRM_REQUIRE(nc >= 0 && (nc == 0 || h_dc), "Incorrect dx/dy/dc size.");
	NRM_REQUIRE(rank >= 0, "Negative dcr detected.");
	NRM_REQUIRE(rank <= nc, "dcr higher than covariate dimension.");
	NRM_REQUIRE(n > (int64_t)rank + dimreduce + 1,
				"Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.");
	NRM_REQUIRE(h_p && h_stat && h_vary, "nrm_association_tests_host: null output");
	const double dof = (double)(n - 1 - rank - dimreduce);
	hipStream_t st = nullptr;
	const int64_t kp = round_up(n, NRM_K_TILE), mp = round_up(nx, NRM_ROW_TILE), np_ = round_up(ny, NRM_ROW_TILE);

	DevBuf cmax, dx, dy, dc, dci, rx, ry, ssx, ssy, bx, by, dot, flags, op, ostat, oalpha, orr, ot;
	// the caller's result arrays are page-locked in place by a helper thread while K1/K2 run (started after the uploads:
	// a hipHostRegister racing a pageable H2D copy was measured to stall that copy by ~20 ms)
	const size_t ob = (size_t)nx * ny * esize(out_dtype);
	HostPin pin_p, pin_s, pin_r, pin_t;
	std::thread pinner;
	Joiner joiner{pinner};
	// covariates as fp64
	std::vector<double> c64;
	if (nc > 0) {
		c64.resize((size_t)nc * n);
		if (c_dtype == NRM_F64)
			memcpy(c64.data(), h_dc, c64.size() * 8);
		else
			for (size_t i = 0; i < c64.size(); i++) c64[i] = ((const float*)h_dc)[i];
		NRM_TRY(dc.alloc(c64.size() * 8));
		NRM_HIP(hipMemcpy(dc.p, c64.data(), c64.size() * 8, hipMemcpyHostToDevice));
		std::vector<double> cm((size_t)nc, 0.0);  // max |C_c| per covariate row: K1's bound on the residuals it quantises
		for (int64_t c = 0; c < nc; c++)
			for (int64_t k = 0; k < n; k++) cm[(size_t)c] = std::max(cm[(size_t)c], std::fabs(c64[(size_t)(c * n + k)]));
		NRM_TRY(cmax.alloc((size_t)nc * 8));
		NRM_HIP(hipMemcpy(cmax.p, cm.data(), (size_t)nc * 8, hipMemcpyHostToDevice));
		NRM_TRY(dci.alloc((size_t)nc * nc * 8));
		NRM_REQUIRE(h_dci != nullptr || rank == 0, "Unmatching dci dimensions.");
		if (h_dci) NRM_HIP(hipMemcpy(dci.p, h_dci, (size_t)nc * nc * 8, hipMemcpyHostToDevice));
	}
	const bool want_alpha = h_alpha != nullptr && nc > 0;
	NRM_REQUIRE(!(want_alpha && samexy), "alpha is not provided for dy == NULL (meaningless in the reference, association.py:1066-1068)");
	DevBuf qx, qy, ex, ey, fx, fy;
	const double guard_tol = guard_tolerance();
	NRM_TRY(dx.alloc((size_t)nx * n * esize(x_dtype)));
	NRM_TRY_RC(nrm_upload(h_dx, dx.p, (int64_t)nx * n * esize(x_dtype), 0, (void*)st));  // (from half a GB up: host threads fill page-locked blocks beside the DMA, nrm_upload.hip)
	// A design matrix with few entries (a CRISPR screen's gRNA incidence): the sparse-design kernels -- the expression rows read once, raw, the
	// contraction replaced by gathers at the design's entries (nrm_host_entries.hip; what normalisr_amd.engine does for the Python host).
	// Same size rule as there; NRM_DE_SPARSE=0 switches it off, =force takes it whatever the size.
	// de with few design rows (case-control DE, BASELINE configs[2]): the raw expression rows streamed once against [C; X~], as the Python engine does
	// (engine.association_de_streaming); NRM_DEBUG de_path=general keeps K1 + K2
	if (allow_sparse && !samexy && nx + nc <= 32 && !de_path_general()) {
		return nrm_host_de_streaming(dx.p, x_dtype, nx, h_dy, y_dtype, ny, c64.data(), nc, n, h_dci, rank, dof, return_dot ? 0 : 1, h_p, h_stat, want_alpha ? h_alpha : nullptr,
									 h_varx, h_vary, h_r, h_t, out_dtype);
	}
	bool dy_up = false;
	if (allow_sparse && !samexy && nc <= nrm_de_sparse_max_covariates()) {
		const char* mode = getenv("NRM_DE_SPARSE");
		const bool off = mode && !strcmp(mode, "0"), force = mode && !strcmp(mode, "force");
		if (!off && (force || (nx >= 32 && ny >= 64 && n >= 2048 && nx * n >= (1ll << 22)))) {
			NRM_TRY(dy.alloc((size_t)ny * n * esize(y_dtype)));
			NRM_TRY_RC(nrm_upload(h_dy, dy.p, (int64_t)ny * n * esize(y_dtype), 0, (void*)st));
			dy_up = true;
			int taken = 0;
			int64_t back = 0;
			NRM_TRY(nrm_host_de_sparse(dx.p, x_dtype, nx, dy.p, y_dtype, ny, dc.as<double>(), c64.data(), nc, n, dci.as<double>(), rank, dof, (return_dot ? 0 : 1), h_p, h_stat,
									   want_alpha ? h_alpha : nullptr, h_varx, h_vary, h_r, h_t, out_dtype, &taken, &back));
			if (taken) {
				if (back > 0 && guard_hits) *guard_hits = back;
				return NRM_OK;
			}
		}
	}
	NRM_TRY(ssx.alloc((size_t)mp * 8));
	if (want_alpha) NRM_TRY(bx.alloc((size_t)nx * nc * 8));
	if (want_alpha) NRM_HIP(hipMemsetAsync(bx.p, 0, (size_t)nx * nc * 8, st));
	if (nslices) {
		NRM_TRY(qx.alloc((size_t)nrm_quant_bytes(mp, kp, nslices)));
		NRM_TRY(ex.alloc((size_t)mp * 4));
		NRM_TRY(fx.alloc((size_t)mp * 8 * 8));
		NRM_TRY(nrm_residualize_q(dx.p, x_dtype, nx, n, n, dc.as<double>(), nc, n, dci.as<double>(), rank, nullptr, kp, mp, ssx.as<double>(),
								  want_alpha ? bx.as<double>() : nullptr, nslices, qx.p, ex.as<int32_t>(), 0, nc ? cmax.as<double>() : nullptr,
								  fx.as<double>(), st));
	} else {
		NRM_TRY(rx.alloc((size_t)mp * kp * 8));
		NRM_TRY(nrm_residualize(dx.p, x_dtype, nx, n, n, dc.as<double>(), nc, n, dci.as<double>(), rank, rx.as<double>(), kp, mp,
								ssx.as<double>(), want_alpha ? bx.as<double>() : nullptr, st));
	}
	if (!samexy) {
		if (!dy_up) {
			NRM_TRY(dy.alloc((size_t)ny * n * esize(y_dtype)));
			NRM_TRY_RC(nrm_upload(h_dy, dy.p, (int64_t)ny * n * esize(y_dtype), 0, (void*)st));
		}
		NRM_TRY(ssy.alloc((size_t)np_ * 8));
		if (want_alpha) NRM_TRY(by.alloc((size_t)ny * nc * 8));
		if (want_alpha) NRM_HIP(hipMemsetAsync(by.p, 0, (size_t)ny * nc * 8, st));
		if (nslices) {
			NRM_TRY(qy.alloc((size_t)nrm_quant_bytes(np_, kp, nslices)));
			NRM_TRY(ey.alloc((size_t)np_ * 4));
			NRM_TRY(fy.alloc((size_t)np_ * 8 * 8));
			NRM_TRY(nrm_residualize_q(dy.p, y_dtype, ny, n, n, dc.as<double>(), nc, n, dci.as<double>(), rank, nullptr, kp, np_, ssy.as<double>(),
									  want_alpha ? by.as<double>() : nullptr, nslices, qy.p, ey.as<int32_t>(), 0, nc ? cmax.as<double>() : nullptr,
									  fy.as<double>(), st));
		} else {
			NRM_TRY(ry.alloc((size_t)np_ * kp * 8));
			NRM_TRY(nrm_residualize(dy.p, y_dtype, ny, n, n, dc.as<double>(), nc, n, dci.as<double>(), rank, ry.as<double>(), kp, np_,
									ssy.as<double>(), want_alpha ? by.as<double>() : nullptr, st));
		}
	}
	const double* A = rx.as<double>();
	const double* B = samexy ? A : ry.as<double>();
	const double* sx = ssx.as<double>();
	const double* sy = samexy ? sx : ssy.as<double>();
	pinner = std::thread([&] {
		pin_p.try_pin(h_p, (int64_t)ob);
		pin_s.try_pin(h_stat, (int64_t)ob);
		pin_r.try_pin(h_r, (int64_t)ob);
		pin_t.try_pin(h_t, (int64_t)ob);
	});
	NRM_TRY(dot.alloc((size_t)mp * np_ * 8));
	DevBuf gwork;
	NRM_TRY(gwork.alloc((size_t)nrm_gram_workspace_bytes()));
	NRM_TRY(flags.alloc(16));
	NRM_HIP(hipMemsetAsync(flags.p, 0, 16, st));
	const double* fxp = nslices ? fx.as<double>() : nullptr;
	const double* fyp = nslices ? (samexy ? fxp : fy.as<double>()) : nullptr;
	NRM_TRY(op.alloc(ob));
	NRM_TRY(ostat.alloc(ob));
	if (h_r) NRM_TRY(orr.alloc(ob));
	if (h_t) NRM_TRY(ot.alloc(ob));
	// coex always converts to covariance (association.py:1037-1039); de keeps gamma unless return_dot
	const int stat_kind = (samexy || return_dot) ? 0 : 1;
	// K2 -> K3 per band of output rows; finished bands are copied out on a second stream while later bands compute
	// (the reference's gather loop, association.py:997-1034, consumes finished tiles the same way)
	const int64_t band = 8 * NRM_ROW_TILE;
	CopyStream cs;
	NRM_HIP(hipStreamCreateWithFlags(&cs.s, hipStreamNonBlocking));
	for (int64_t a = 0; a < nx; a += band) {
		const int64_t b = std::min(nx, a + band);
		if (nslices)
			NRM_TRY(nrm_gram_i8_band(qx.p, ex.as<int32_t>(), 0, samexy ? qx.p : qy.p, samexy ? ex.as<int32_t>() : ey.as<int32_t>(), 0, mp, np_, kp, nslices,
									 dot.as<double>(), np_, samexy ? 1 : 0, nx, ny, a, b == nx ? mp : b, gwork.p, st));
		else
			NRM_TRY(nrm_gram_f64_band(A, B, mp, np_, kp, kp, kp, dot.as<double>(), np_, samexy ? 1 : 0, nx, ny, a, b == nx ? mp : b, gwork.p, st));
		NRM_TRY(nrm_assoc_sweep_band(dot.as<double>(), np_, sx, sy, nx, ny, n, dof, samexy ? 1 : 0, stat_kind, op.p, ostat.p,
									 h_r ? orr.p : nullptr, h_t ? ot.p : nullptr, out_dtype, ny, flags.as<int32_t>(), a, b, nslices, fxp, fyp, guard_tol, st));
		hipEvent_t ev;
		NRM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		cs.events.push_back(ev);
		NRM_HIP(hipEventRecord(ev, st));
	}
	if (pinner.joinable()) pinner.join();
	const size_t row = (size_t)ny * esize(out_dtype);
	for (int64_t a = 0, i = 0; a < nx; a += band, i++) {
		const int64_t b = std::min(nx, a + band);
		NRM_HIP(hipStreamWaitEvent(cs.s, cs.events[(size_t)i], 0));
		const size_t off = (size_t)a * row, len = (size_t)(b - a) * row;
		NRM_HIP(hipMemcpyAsync((char*)h_p + off, (const char*)op.p + off, len, hipMemcpyDeviceToHost, cs.s));
		NRM_HIP(hipMemcpyAsync((char*)h_stat + off, (const char*)ostat.p + off, len, hipMemcpyDeviceToHost, cs.s));
		if (h_r) NRM_HIP(hipMemcpyAsync((char*)h_r + off, (const char*)orr.p + off, len, hipMemcpyDeviceToHost, cs.s));
		if (h_t) NRM_HIP(hipMemcpyAsync((char*)h_t + off, (const char*)ot.p + off, len, hipMemcpyDeviceToHost, cs.s));
	}
	if (want_alpha) {
		// alpha comes from gamma whatever return_dot says (association.py:238-243 vs :1044-1048)
		NRM_TRY(oalpha.alloc(ob * nc));
		NRM_TRY(nrm_alpha(ostat.p, out_dtype, ny, stat_kind, ssx.as<double>(), n, bx.as<double>(), by.as<double>(), nx, ny, nc, oalpha.p,
						  out_dtype, st));
	}
	NRM_HIP(hipStreamSynchronize(st));
	NRM_HIP(hipStreamSynchronize(cs.s));
	int32_t hf[4];
	NRM_HIP(hipMemcpy(hf, flags.p, 16, hipMemcpyDeviceToHost));
	if (hf[0] || hf[1]) {
		nrm_set_error("association results failed the reference's assertions (association.py:248,252): %d tiles non-finite, %d tiles with R^2 > 1+1e-8", hf[0], hf[1]);
		return NRM_E_NUMERIC;
	}
	if (nslices) {
		float w;
		memcpy(&w, &hf[3], 4);
		g_guard_worst = (double)w;
		if (guard_hits) *guard_hits = hf[2];
		if (hf[2] > 0 && guard_hits) return NRM_OK;  // the caller redoes the call on the fp64 kernel: nothing more to bring back from this pass
	}
	if (want_alpha) NRM_HIP(hipMemcpy(h_alpha, oalpha.p, ob * nc, hipMemcpyDeviceToHost));
	// variances = ss / n with the 0 -> 1 rule (association.py:230-233), cast to the output dtype
	std::vector<double> hs((size_t)std::max(mp, np_));
	auto emit_var = [&](const double* d_ss, int64_t cnt, void* h_out) -> int {
		NRM_HIP(hipMemcpy(hs.data(), d_ss, (size_t)cnt * 8, hipMemcpyDeviceToHost));
		for (int64_t i = 0; i < cnt; i++) {
			double v = hs[i] / (double)n;
			if (v == 0.0) v = 1.0;
			if (out_dtype == NRM_F64)
				((double*)h_out)[i] = v;
			else
				((float*)h_out)[i] = (float)v;
		}
		return NRM_OK;
	};
	NRM_TRY(emit_var(sy, ny, h_vary));
	if (h_varx && !samexy) NRM_TRY(emit_var(sx, nx, h_varx));
	return NRM_OK;
}
