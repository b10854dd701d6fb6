// What the whole-problem host entries (numpy buffers in, numpy buffers out; no torch) share: the device scratch pool kept between calls, the
// lock that serialises them per process, a scoped device buffer.  nrm_api.hip owns the pool and the lock; nrm_host_entries.hip (the sparse-design
// path of nrm_association_tests_host, the single=1 / single=4 / binnet entries) uses them.
#pragma once
#include <mutex>
#include <thread>
#include "nrm_common.h"
#include "nrm_host_logic.h"

struct NrmHipAlloc {
	void* alloc(size_t bytes) {
		void* p = nullptr;
		if (hipMalloc(&p, bytes) != hipSuccess) {
			(void)hipGetLastError();
			return nullptr;
		}
		return p;
	}
	void free(void* p) { (void)hipFree(p); }
};
typedef DevPoolT<NrmHipAlloc> NrmDevPool;
NrmDevPool& nrm_host_pool();     // (nrm_api.hip)
std::mutex& nrm_host_entry_mutex();  // one whole-problem call at a time per process (the pool and the default stream are shared)
int nrm_bind_device(void);              // the calling thread onto the device nrm_set_device chose for the process (nrm_api.hip); nothing when none was chosen

struct DevBuf {
	void* p = nullptr;
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	~DevBuf() { release(); }
	void release() {
		if (p) {
			(void)hipDeviceSynchronize();
			nrm_host_pool().give(p);
			p = nullptr;
		}
	}
	int alloc(size_t bytes) {
		release();
		p = nrm_host_pool().take(bytes ? bytes : 16);
		if (!p) {
			nrm_set_error("hipMalloc of %zu bytes failed", bytes);
			return NRM_E_DEVICE;
		}
		return NRM_OK;
	}
	template <typename T>
	T* as() const {
		return reinterpret_cast<T*>(p);
	}
};

// Page-lock of a caller-owned result array for the duration of one call (a pageable device-to-host copy runs at a tenth of the PCIe rate); a
// range that cannot be locked (already registered by the caller, locked-memory limit) is simply copied to at the pageable rate.
struct NrmHostPin {
	void* p = nullptr;
	void try_pin(void* q, int64_t bytes) {
		if (q && bytes >= (1 << 20) && nrm_host_pin(q, bytes, 0) == NRM_OK) p = q;
	}
	~NrmHostPin() {
		if (p) {
			(void)hipDeviceSynchronize();
			(void)hipHostUnregister(p);
		}
	}
};

#define NRM_TRY(call)        \
	do {                     \
		int rc_ = (call);    \
		if (rc_) return rc_; \
	} while (0)

static inline int64_t nrm_round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
static inline size_t nrm_esize(int dtype) { return dtype == NRM_F64 ? 8 : 4; }

// The design matrix (already in HBM) as the lists of csrc/nrm_design_lists.hip: CSR always, ELL on request.  ok: 0 < entries <= max_density nx n.
struct NrmDesignLists {
	DevBuf cnt, coff, info, row_ptr, slot2x, sig, pos, w, base, cells, row_vals, ell, ellv;
	int64_t nnz = 0, padded = 0, nslots = 0, ngroups = 0, nch = 0;
	int bits = 0;
	bool binary = true, ok = false;
	int build(const void* d_x, int x_dtype, int64_t nx, int64_t n, bool want_ell, double max_density, hipStream_t st);
};

// the sparse-design form of single=0 de inside nrm_association_tests_host (nrm_host_entries.hip); *taken = 0: the design does not qualify
// (dense, or empty) and nothing was written; *handed_back = rows too close to the span of the covariates (the caller runs the dense fp64 path)
int nrm_host_de_sparse(const void* d_x, int x_dtype, int64_t nx, const void* d_y, int y_dtype, int64_t ny, const double* d_c, const double* h_c64, int64_t nc, int64_t n,
					   const double* d_dci, int rank, double dof, int stat_kind, void* h_p, void* h_stat, void* h_alpha, void* h_varx, void* h_vary, void* h_r, void* h_t,
					   int out_dtype, int* taken, int64_t* handed_back);
// de with nx + nc <= 32: the streaming kernel (nrm_host_entries.hip); d_x in HBM, h_dy uploaded inside
int nrm_host_de_streaming(const void* d_x, int x_dtype, int64_t nx, const void* h_dy, int y_dtype, int64_t ny, const double* h_c64, int64_t nc, int64_t n,
						  const double* h_dci, int rank, double dof, int stat_kind, void* h_p, void* h_stat, void* h_alpha, void* h_varx, void* h_vary, void* h_r, void* h_t,
						  int out_dtype);
