// Shared host-side helpers for libnormalisr_hip.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../include/normalisr_hip.h"

void nrm_set_error(const char* fmt, ...);

#define NRM_HIP(call)                                                                      \
	do {                                                                                   \
		hipError_t e_ = (call);                                                            \
		if (e_ != hipSuccess) {                                                            \
			(void)hipGetLastError(); /* the error is reported here: do not leave it for the next launch check */ \
			nrm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
						  __LINE__);                                                       \
			return NRM_E_DEVICE;                                                           \
		}                                                                                  \
	} while (0)

#define NRM_REQUIRE(cond, ...)      \
	do {                            \
		if (!(cond)) {              \
			nrm_set_error(__VA_ARGS__); \
			return NRM_E_ARG;       \
		}                           \
	} while (0)

#define NRM_TRY_RC(call)     \
	do {                     \
		int rc_ = (call);    \
		if (rc_) return rc_; \
	} while (0)

static inline int nrm_check_launch(const char* what) {
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		nrm_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
		return NRM_E_DEVICE;
	}
	return NRM_OK;
}

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));
