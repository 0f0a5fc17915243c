"""What the drop-in's host-to-device upload costs piece by piece (hipHostRegister / copy / unregister of a 5.24-GB user buffer), and
whether pinning and copying in pieces -- piece i + 1 is registered while piece i travels -- hides the registration.
usage (GPU box): python tools/probes/upload_probe.py [GB]"""
import ctypes as C, sys, time
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 5.24
N = int(GB * 1e9) // 4096 * 4096
x = np.ones(N // 4, np.float32)           # pages touched
d = torch.empty(N // 4, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
vp, sz = C.c_void_p, C.c_size_t
def T(f):
    t0 = time.perf_counter(); r = f(); return (time.perf_counter() - t0) * 1e3, r
for rep in range(2):
    tr, rc = T(lambda: hip.hipHostRegister(vp(x.ctypes.data), sz(N), 0))
    tc, _ = T(lambda: (hip.hipMemcpy(vp(d.data_ptr()), vp(x.ctypes.data), sz(N), 1), hip.hipDeviceSynchronize()))
    tu, _ = T(lambda: hip.hipHostUnregister(vp(x.ctypes.data)))
    print(f"whole buffer {N/1e9:.2f} GB: register {tr:.1f} ms (rc {rc}), copy {tc:.1f} ms ({N/tc/1e6:.1f} GB/s), unregister {tu:.1f} ms, sum {tr+tc+tu:.1f} ms")
s = C.c_void_p()
hip.hipStreamCreate(C.byref(s))
for piece_mb in (64, 128, 256, 512, 1024):
    P = piece_mb << 20
    for rep in range(2):
        t0 = time.perf_counter()
        off = 0
        hip.hipHostRegister(vp(x.ctypes.data), sz(min(P, N)), 0)
        while off < N:
            n = min(P, N - off)
            hip.hipMemcpyAsync(vp(d.data_ptr() + off), vp(x.ctypes.data + off), sz(n), 1, s)
            nxt = off + n
            if nxt < N:
                hip.hipHostRegister(vp(x.ctypes.data + nxt), sz(min(P, N - nxt)), 0)   # while the piece above travels
            off = nxt
        hip.hipStreamSynchronize(s)
        t1 = time.perf_counter()
        off = 0
        while off < N:
            hip.hipHostUnregister(vp(x.ctypes.data + off)); off += P
        t2 = time.perf_counter()
    print(f"pieces of {piece_mb} MB: upload {1e3*(t1-t0):.1f} ms ({N/(t1-t0)/1e9:.1f} GB/s), unregister {1e3*(t2-t1):.1f} ms, sum {1e3*(t2-t0):.1f} ms")

# FIRST upload of a buffer (what a one-call process such as the command line pays): fresh pages every time
def fresh():
    return np.ones(N // 4, np.float32)
for mode in ("whole", 64, 256, 1024, "pageable"):
    y = fresh()
    t0 = time.perf_counter()
    if mode == "whole":
        hip.hipHostRegister(vp(y.ctypes.data), sz(N), 0)
        hip.hipMemcpyAsync(vp(d.data_ptr()), vp(y.ctypes.data), sz(N), 1, s); hip.hipStreamSynchronize(s)
    elif mode == "pageable":
        hip.hipMemcpy(vp(d.data_ptr()), vp(y.ctypes.data), sz(N), 1); hip.hipDeviceSynchronize()
    else:
        P = mode << 20
        off = 0
        while off < N:
            n = min(P, N - off)
            hip.hipHostRegister(vp(y.ctypes.data + off), sz(n), 0)
            hip.hipMemcpyAsync(vp(d.data_ptr() + off), vp(y.ctypes.data + off), sz(n), 1, s)
            off += n
        hip.hipStreamSynchronize(s)
    t1 = time.perf_counter()
    if mode == "whole":
        hip.hipHostUnregister(vp(y.ctypes.data))
    elif mode != "pageable":
        off = 0
        while off < N:
            hip.hipHostUnregister(vp(y.ctypes.data + off)); off += mode << 20
    print(f"first upload of fresh pages, {mode}: {1e3*(t1-t0):.1f} ms ({N/(t1-t0)/1e9:.1f} GB/s)")
    del y

# The library's own two uploads of a 10 000 x 131 072 ensemble (round 5): tspws_hip_upload (one device) against
# tspws_hip_multi_upload (a device list: here virtual shards of the one GPU, so the link is shared -- the point is that the pieces of
# the several-device form add nothing in front of the copies any more)
import importlib, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import abi
tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
mtr, Ns = 10000, 131072
del x, d
torch.cuda.empty_cache()
H = np.ones((mtr, Ns), np.float32)
dd = torch.empty((mtr, Ns), dtype=torch.float32, device="cuda")
for rep in range(3):
    t0 = time.perf_counter()
    assert lib.tspws_hip_upload(C.c_void_p(dd.data_ptr()), C.c_void_p(H.ctypes.data), C.c_size_t(H.nbytes), None) == 0
    t1 = time.perf_counter()
    print(f"tspws_hip_upload, {H.nbytes/1e9:.2f} GB: {1e3*(t1-t0):.1f} ms ({H.nbytes/(t1-t0)/1e9:.1f} GB/s)")
del dd
torch.cuda.empty_cache()
p = tspws.resolve(abi.default_params(Kmax=10, unbiased=1), Ns)
for ndev in (2, 4):
    m = C.c_void_p()
    arr = (C.c_int * ndev)(*([0] * ndev))
    tspws.check(lib.tspws_hip_multi_create(C.byref(m), ndev, arr, p.type, p.J, p.V, Ns, p.s0, p.b0, p.w0, int(p.uni)), "multi_create")
    sh, dl, dt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for rep in range(3):
        t0 = time.perf_counter()
        tspws.check(lib.tspws_hip_multi_upload(m, H.ctypes.data, Ns, mtr, C.byref(sh), C.byref(dl), C.byref(dt)), "multi_upload")
        t1 = time.perf_counter()
        print(f"tspws_hip_multi_upload, {ndev} virtual shards: {1e3*(t1-t0):.1f} ms ({H.nbytes/(t1-t0)/1e9:.1f} GB/s)")
    lib.tspws_hip_multi_destroy(m)
