import time, torch, numpy as np
n = 1 << 30
a = np.ones(n, dtype=np.uint8)
d = torch.empty(n, dtype=torch.uint8, device='cuda')
for name, src in [('pageable', torch.from_numpy(a)), ('pinned', torch.from_numpy(a).pin_memory())]:
    d[:1 << 20].copy_(src[:1 << 20]); torch.cuda.synchronize()
    t = time.perf_counter(); d.copy_(src, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(name, 'H2D %.1f GB/s' % (n / dt / 1e9))
t = time.perf_counter(); p = torch.from_numpy(a).pin_memory(); print('pin_memory (alloc+copy) %.2f s' % (time.perf_counter() - t))
t = time.perf_counter(); torch.cuda.cudart().cudaHostRegister(a.ctypes.data, n, 0); print('hostRegister in place %.2f s' % (time.perf_counter() - t))
t = time.perf_counter(); d.copy_(torch.from_numpy(a), non_blocking=True); torch.cuda.synchronize(); print('registered H2D %.1f GB/s' % (n / (time.perf_counter() - t) / 1e9))
