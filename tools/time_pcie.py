import numpy as np, torch, time
a = np.random.default_rng(0).standard_normal((5000, 10000), dtype=np.float32)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d = torch.from_numpy(a).cuda(); torch.cuda.synchronize(); t1 = time.perf_counter()
print('pageable H2D 200MB: %.2f ms (%.1f GB/s)' % ((t1 - t0) * 1e3, a.nbytes / (t1 - t0) / 1e9))
pin = torch.empty(a.shape, dtype=torch.float32, pin_memory=True)
for rep in range(3):
    t0 = time.perf_counter(); pin.copy_(torch.from_numpy(a)); t1 = time.perf_counter(); d.copy_(pin, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host copy to pinned: %.2f ms; pinned H2D: %.2f ms (%.1f GB/s)' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, a.nbytes / (t2 - t1) / 1e9))
out = torch.empty(a.shape, dtype=torch.float32, pin_memory=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out.copy_(d, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
print('pinned D2H 200MB: %.2f ms (%.1f GB/s)' % ((t1 - t0) * 1e3, a.nbytes / (t1 - t0) / 1e9))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); h = d.cpu(); t1 = time.perf_counter()
print('pageable D2H 200MB: %.2f ms' % ((t1 - t0) * 1e3))
t0 = time.perf_counter(); x = torch.empty((5000, 5000), dtype=torch.float32, pin_memory=True); t1 = time.perf_counter()
print('pinned alloc 100MB (cached allocator, 2nd+ call cheap): %.2f ms' % ((t1 - t0) * 1e3))
