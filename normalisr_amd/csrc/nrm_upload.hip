// Host -> device copies of the caller's (pageable) arrays.  torch's / HIP's pageable copy of the 3 GB expression matrix of BASELINE
// configs[3] runs at 37 GB/s on this platform; page-locking the caller's array first costs 25 ms for the same 3 GB and then copies at
// 57 GB/s -- no faster in sum.  Here: a ring of page-locked staging blocks owned by the library; host threads copy the next block of the
// source into the ring while the previous block's DMA runs (32 MB blocks: 3 GB in 55 ms = 54 GB/s; 200 MB in 4.8 ms, where the runtime's
// pageable copy takes 3.6 -- callers use this from half a GB up), so the call runs at the DMA's rate and the caller's array is never
// page-locked (and may be reused the moment the call returns: what is still in flight comes from the ring).
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "nrm_common.h"

namespace {

constexpr int64_t UP_BLOCK_MAX = 32ll << 20;  // bytes per staging block (4 blocks = 128 MB page-locked once, ~10 ms; NRM_UPLOAD_BLOCK_MB uses less of a block)
constexpr int UP_SLOTS = 4;

struct Ring {
	void* slot[UP_SLOTS] = {};
	hipEvent_t done[UP_SLOTS] = {};
	bool used[UP_SLOTS] = {};
	int device = -1;
	std::mutex lock;  // one upload at a time per process through the ring
	~Ring() { release(); }
	void release() {
		for (int i = 0; i < UP_SLOTS; i++) {
			if (done[i]) (void)hipEventDestroy(done[i]);
			if (slot[i]) (void)hipHostFree(slot[i]);
			slot[i] = nullptr;
			done[i] = nullptr;
			used[i] = false;
		}
		device = -1;
	}
};
Ring g_ring;

void copy_parallel(char* dst, const char* src, int64_t bytes, int threads) {
	if (threads <= 1 || bytes < (8 << 20)) {
		memcpy(dst, src, (size_t)bytes);
		return;
	}
	const int64_t per = ((bytes + threads - 1) / threads + 4095) / 4096 * 4096;
	std::vector<std::thread> th;
	for (int t = 1; t < threads; t++) {
		const int64_t lo = per * t, hi = lo + per < bytes ? lo + per : bytes;
		if (hi > lo) th.emplace_back([=] { memcpy(dst + lo, src + lo, (size_t)(hi - lo)); });
	}
	memcpy(dst, src, (size_t)(per < bytes ? per : bytes));
	for (auto& x : th) x.join();
}

}  // namespace

// h_src (pageable or not) -> d_dst, `bytes` bytes, on `stream`.  Returns when the last block has been handed to the DMA engine (the copy
// itself completes in stream order); threads: host threads per block, 0 = choose.
extern "C" int nrm_upload(const void* h_src, void* d_dst, int64_t bytes, int threads, void* stream) {
	NRM_REQUIRE(bytes >= 0, "nrm_upload: negative size");
	if (bytes == 0) return NRM_OK;
	NRM_REQUIRE(h_src && d_dst, "nrm_upload: null pointer");
	hipStream_t st = (hipStream_t)stream;
	if (bytes < (512ll << 20) && !getenv("NRM_UPLOAD_BLOCK_MB")) {  // below half a GB the runtime's own pageable copy is the faster one (200 MB: 3.6 against 4.8 ms)
		NRM_HIP(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, st));
		return NRM_OK;
	}
	if (threads <= 0) {  // (measured: 4 threads fill a 32 MB block in ~0.5 ms, faster than its DMA; 16 threads cost more to start than they save)
		threads = (int)std::thread::hardware_concurrency() / 2;
		threads = threads < 1 ? 1 : (threads > 4 ? 4 : threads);
	}
	std::lock_guard<std::mutex> guard(g_ring.lock);
	int dev = 0;
	NRM_HIP(hipGetDevice(&dev));
	if (g_ring.device != dev) {  // (page-locked memory is mapped for the device current at allocation)
		g_ring.release();
		for (int i = 0; i < UP_SLOTS; i++) {
			NRM_HIP(hipHostMalloc(&g_ring.slot[i], (size_t)UP_BLOCK_MAX, hipHostMallocDefault));
			NRM_HIP(hipEventCreateWithFlags(&g_ring.done[i], hipEventDisableTiming));
		}
		g_ring.device = dev;
	}
	int64_t UP_BLOCK = 32ll << 20;
	if (const char* e = getenv("NRM_UPLOAD_BLOCK_MB")) UP_BLOCK = (int64_t)atoi(e) << 20;
	if (UP_BLOCK < (1 << 20) || UP_BLOCK > UP_BLOCK_MAX) UP_BLOCK = UP_BLOCK_MAX;
	if (const char* e = getenv("NRM_UPLOAD_THREADS")) threads = atoi(e) > 0 ? atoi(e) : threads;
	const char* src = (const char*)h_src;
	char* dst = (char*)d_dst;
	int i = 0;
	for (int64_t off = 0; off < bytes; off += UP_BLOCK, i = (i + 1) % UP_SLOTS) {
		const int64_t len = bytes - off < UP_BLOCK ? bytes - off : UP_BLOCK;
		if (g_ring.used[i]) NRM_HIP(hipEventSynchronize(g_ring.done[i]));  // the DMA that last read this block is through
		copy_parallel((char*)g_ring.slot[i], src + off, len, threads);
		NRM_HIP(hipMemcpyAsync(dst + off, g_ring.slot[i], (size_t)len, hipMemcpyHostToDevice, st));
		NRM_HIP(hipEventRecord(g_ring.done[i], st));
		g_ring.used[i] = true;
	}
	return NRM_OK;
}

extern "C" int nrm_upload_release(void) {
	std::lock_guard<std::mutex> guard(g_ring.lock);
	if (g_ring.device >= 0) (void)hipDeviceSynchronize();
	g_ring.release();
	return NRM_OK;
}
