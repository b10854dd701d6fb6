// Where does K3's time go?  The P-value function of csrc/nrm_pvalue.h on 12.5M R^2 values of the C2 kind (null pairs: r ~ N(0, 1/n)),
// piece by piece: each variant adds one stage of the fast path.  16 values per thread, unrolled, as in k_assoc_sweep_sym.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/pvalue_probe tools/pvalue_probe.hip && ./tools/pvalue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "../normalisr_amd/csrc/nrm_host_logic.h"
#include "../normalisr_amd/csrc/nrm_pvalue.h"

template <int MODE>
__device__ __forceinline__ double stage(double r2, const PvalPlan& pl) {
	if (MODE == 0) return r2;
	double x = 1.0 - r2;
	double w = 1.0 - x;
	double u = -log1p(-w);
	if (MODE == 1) return u;
	double z = pl.alpha * u;
	double sz = sqrt(z);
	if (MODE == 2) return sz;
	double poly = pl.coef[NRM_PCOEF - 1];
#pragma unroll
	for (int j = NRM_PCOEF - 2; j >= 0; j--) poly = fma(poly, u, pl.coef[j]);
	if (MODE == 3) return sz * poly;
	double e = exp(-z);
	if (MODE == 4) return e * (sz * poly);
	if (MODE == 5) return e * (erfcx(sz) + sz * poly);
	if (MODE == 7) return erfc(sz) + e * (sz * poly);
	return nrm_pvalue(r2, pl);
}

template <int MODE>
__global__ void __launch_bounds__(256, 3) k_probe(const double* __restrict__ r2, int64_t count, PvalPlan pl, float* __restrict__ out) {
	const int64_t base = (int64_t)blockIdx.x * 4096 + threadIdx.x;
	float pv[16];
#pragma unroll
	for (int i = 0; i < 16; i++) {
		const int64_t k = base + i * 256;
		pv[i] = k < count ? (float)stage<MODE>(r2[k], pl) : 0.f;
	}
#pragma unroll
	for (int i = 0; i < 16; i++) {
		const int64_t k = base + i * 256;
		if (k < count) out[k] = pv[i];
	}
}

int main() {
	const int64_t count = 12497500, n = 10000;
	std::vector<double> h(count);
	std::mt19937_64 g(5);
	std::normal_distribution<double> nd(0.0, 1.0 / std::sqrt((double)n));
	for (auto& v : h) {
		double r = nd(g);
		v = r * r;
	}
	double* d;
	float* o;
	if (hipMalloc(&d, count * 8) != hipSuccess || hipMalloc(&o, count * 4) != hipSuccess) return 1;
	(void)hipMemcpy(d, h.data(), count * 8, hipMemcpyHostToDevice);
	nrm_pvalue_plan plan;
	nrm_pvalue_plan_init_host(&plan, (double)(n - 4));
	PvalPlan pl;
	pl.a = plan.a; pl.alpha = plan.alpha; pl.ln_front = plan.ln_front; pl.umax = plan.umax;
	for (int j = 0; j < NRM_PCOEF; j++) pl.coef[j] = plan.coef[j];
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	const int grid = (int)((count + 4095) / 4096);
	auto time = [&](auto kern, const char* what) {
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, count, pl, o);
		(void)hipEventRecord(e0);
		for (int r = 0; r < 10; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, count, pl, o);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		float ms = 0;
		(void)hipEventElapsedTime(&ms, e0, e1);
		printf("%-44s %.4f ms\n", what, ms / 10);
	};
	time(k_probe<0>, "load/store only");
	time(k_probe<1>, "+ log1p");
	time(k_probe<2>, "+ sqrt");
	time(k_probe<3>, "+ 20-term polynomial");
	time(k_probe<4>, "+ exp");
	time(k_probe<5>, "+ erfcx (the whole fast path)");
	time(k_probe<7>, "erfc instead of exp * erfcx");
	time(k_probe<6>, "nrm_pvalue (with the general path behind it)");
	return 0;
}
#include <cstdarg>
void nrm_set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vfprintf(stderr, fmt, ap);
	va_end(ap);
}
