import torch, time, numpy as np, ctypes as C
hip = C.CDLL("libamdhip64.so")
N = 2*1024**3
x = np.ones(N//4, np.float32)
d = torch.empty(N//4, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for rep in range(2):
    t0=time.perf_counter(); d.copy_(torch.from_numpy(x)); torch.cuda.synchronize(); t1=time.perf_counter()
    print("pageable copy 2GB: %.1f ms (%.1f GB/s)"%((t1-t0)*1e3, N/(t1-t0)/1e9))
t0=time.perf_counter(); rc=hip.hipHostRegister(C.c_void_p(x.ctypes.data), C.c_size_t(N), 0); t1=time.perf_counter()
print("hipHostRegister rc",rc,"%.1f ms"%((t1-t0)*1e3))
for rep in range(2):
    t0=time.perf_counter(); rc=hip.hipMemcpy(C.c_void_p(d.data_ptr()), C.c_void_p(x.ctypes.data), C.c_size_t(N), 1); torch.cuda.synchronize(); t1=time.perf_counter()
    print("registered copy 2GB: rc %d %.1f ms (%.1f GB/s)"%(rc,(t1-t0)*1e3, N/(t1-t0)/1e9))
t0=time.perf_counter(); hip.hipHostUnregister(C.c_void_p(x.ctypes.data)); t1=time.perf_counter()
print("unregister %.1f ms"%((t1-t0)*1e3))
