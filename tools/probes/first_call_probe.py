"""Where the first tspws_main call of a process spends its ~250 ms (no torch in the process: like the command line).
usage (GPU box): python tools/probes/first_call_probe.py"""
import ctypes as C, os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "tests"))
t0 = time.perf_counter()
lib = C.CDLL(os.environ.get("TSPWS_LIB_PATH", os.path.join(R, "ts-pws_amd", "lib", "libtspws_hip.so")))
t1 = time.perf_counter(); print(f"dlopen libtspws_hip.so (+ libamdhip64): {1e3*(t1-t0):.1f} ms")
lib.tspws_hip_device_count.restype = C.c_int
n = lib.tspws_hip_device_count()
t2 = time.perf_counter(); print(f"device_count = {n}: {1e3*(t2-t1):.1f} ms")
d = C.c_void_p()
rc = lib.tspws_hip_alloc(C.byref(d), C.c_size_t(1 << 20), 0)
t3 = time.perf_counter(); print(f"first hipMalloc (context): rc {rc} {1e3*(t3-t2):.1f} ms")
rc = lib.tspws_hip_zero(d, C.c_size_t(1 << 20), None); lib.tspws_hip_sync(None)
t4 = time.perf_counter(); print(f"first memset + sync: rc {rc} {1e3*(t4-t3):.1f} ms")
import numpy as np, abi
p = abi.default_params()
lib.tspws_resolve_params(C.byref(p), 16501, C.c_float(1.0))
h = C.c_void_p()
lib.tspws_hip_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
t5 = time.perf_counter()
rc = lib.tspws_hip_plan_create(C.byref(h), p.type, p.J, p.V, 16501, p.s0, p.b0, p.w0, int(p.uni), 0)
t6 = time.perf_counter(); print(f"plan_create (first kernel launch: code object load; taps): rc {rc} {1e3*(t6-t5):.1f} ms")
h2 = C.c_void_p()
rc = lib.tspws_hip_plan_create(C.byref(h2), p.type, p.J, p.V, 16501, p.s0, p.b0, p.w0, int(p.uni), 0)
t7 = time.perf_counter(); print(f"second plan_create: rc {rc} {1e3*(t7-t6):.1f} ms")
