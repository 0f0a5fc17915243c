"""ts-pws_amd -- MI355X-native time-scale phase-weighted stack (ts-PWS).

The product is the C-ABI shared library ``lib/libtspws_hip.so`` (hand-written
gfx950 HIP kernels + the C host entry point ``tspws_main``; headers in
``include/``).  This module is only the Python-side binding used by the tests
and ``bench.py``: ctypes signatures, a ``Plan`` wrapper, and helpers that run
the device-resident path on torch-allocated HBM buffers (torch provides device
memory, streams and torch.distributed -- plumbing, not compute).

There is no CPU fallback anywhere: if the library is missing or no HIP device
is present the calls raise.

Import with ``importlib.import_module("ts-pws_amd")`` (the directory name is
not a Python identifier).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("TSPWS_LIB_PATH") or os.path.join(_HERE, "lib", "libtspws_hip.so")  # override: profiling builds only

time_t = C.c_long


class t_tsPWS(C.Structure):
    """include/ts_pws1f_lib.h (reference: src/ts_pws1f_lib.h:19-57)."""
    _fields_ = [
        ("type", C.c_int), ("uni", C.c_uint), ("J", C.c_uint), ("V", C.c_uint),
        ("s0", C.c_double), ("b0", C.c_double), ("w0", C.c_double), ("wu", C.c_double),
        ("fmin", C.c_double), ("Q", C.c_double), ("cycle", C.c_double),
        ("w0set", C.c_int), ("lrm", C.c_int), ("bin", C.c_int), ("lkinst", C.c_int),
        ("lVfix", C.c_int), ("ls0fix", C.c_int), ("lb0fix", C.c_int), ("verbose", C.c_int),
        ("fold", C.c_int), ("unbiased", C.c_int), ("convergence", C.c_int),
        ("subsmpl_N", C.c_uint), ("subsmpl_p", C.c_double),
        ("jackknife_n", C.c_uint), ("jackknife_d", C.c_uint), ("obin", C.c_uint),
        ("AllSteps", C.c_int), ("Nmax", C.c_uint), ("Kmax", C.c_uint),
        ("kinst", C.c_char_p), ("filein", C.c_char_p), ("fileout", C.c_char_p), ("fileconv", C.c_char_p),
    ]


class FrameInfo(C.Structure):
    """tspws_hip_frame_info (include/tspws_hip.h)."""
    _fields_ = [
        ("type", C.c_int), ("S", C.c_uint), ("V", C.c_uint), ("J", C.c_uint), ("N", C.c_uint),
        ("s0", C.c_double), ("b0", C.c_double), ("w0", C.c_double), ("Cpsi", C.c_double),
        ("ncoef", C.c_size_t), ("ntaps", C.c_size_t), ("device", C.c_int),
    ]


class TspwsError(RuntimeError):
    pass


def build(verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if verbose or out.returncode:
        print(out.stdout + out.stderr)
    if out.returncode:
        raise TspwsError("building libtspws_hip.so failed")
    return LIB_PATH


_lib = None

# name -> (restype, argtypes); every symbol include/tspws_hip.h + ts_pws1f_lib.h declare
_vp, _u, _d, _sz, _i, _f = C.c_void_p, C.c_uint, C.c_double, C.c_size_t, C.c_int, C.c_float
SYMBOLS = {
    "tspws_main": (_i, [_vp, _vp, _vp]),
    "tspws_hip_device_count": (_i, []),
    "tspws_hip_last_error": (C.c_char_p, []),
    "tspws_hip_alloc": (_i, [C.POINTER(_vp), _sz, _i]),
    "tspws_hip_free": (_i, [_vp]),
    "tspws_hip_upload": (_i, [_vp, _vp, _sz, _vp]),
    "tspws_hip_download": (_i, [_vp, _vp, _sz, _vp]),
    "tspws_hip_zero": (_i, [_vp, _sz, _vp]),
    "tspws_hip_sync": (_i, [_vp]),
    "tspws_resolve_params": (None, [_vp, _u, _f]),
    "tspws_hip_plan_create": (_i, [C.POINTER(_vp), _i, _u, _u, _u, _d, _d, _d, _i, _i]),
    "tspws_hip_plan_destroy": (None, [_vp]),
    "tspws_hip_plan_info": (_i, [_vp, _vp]),
    "tspws_hip_plan_tables": (_i, [_vp] + [_vp] * 6),
    "tspws_hip_plan_taps": (_i, [_vp, _vp, _vp]),
    "tspws_hip_fold": (_i, [_vp, _sz, _sz, _sz, _vp]),
    "tspws_hip_remove_mean": (_i, [_vp, _sz, _sz, _sz, _vp]),
    "tspws_hip_partial_stacks": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _u, _vp, _sz, _vp]),
    "tspws_hip_partial_stacks_range": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _u, _u, _u, _vp, _sz, _vp]),
    "tspws_hip_forward_f64": (_i, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "tspws_hip_forward_f32": (_i, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "tspws_hip_spectral_first_scale": (_u, [_vp, _u]),
    "tspws_hip_spectral_end_scale": (_u, [_vp]),
    "tspws_hip_spectral_transform_length": (_u, [_vp]),
    "tspws_hip_spectral_choice": (_u, [_vp, _sz]),
    "tspws_hip_forward_spectral_f64": (_i, [_vp, _vp, _sz, _sz, _vp, _u, _vp]),
    "tspws_hip_forward_spectral_f32": (_i, [_vp, _vp, _sz, _sz, _vp, _u, _vp]),
    "tspws_hip_inverse": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "tspws_hip_accumulate": (_i, [_vp, _vp, _sz, _vp, _vp, _i, _vp]),
    "tspws_hip_stacks_double": (_i, [_vp, _vp, _u, _sz, _vp, _vp, _vp]),
    "tspws_hip_stacks_float": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "tspws_hip_weight": (_i, [_vp, _vp, _vp, _vp, _u, _u, _d, _i, _vp]),
    "tspws_hip_epilogue": (_i, [_vp, _vp, _vp, _vp, _sz, _u, _vp]),
    "tspws_hip_stack_local": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "tspws_hip_reduce_buffer": (_i, [_vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz)]),
    "tspws_hip_stack_finish": (_i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "tspws_hip_stack_finish_range": (_i, [_vp, _vp, _sz, _u, _u, _vp]),
    "tspws_hip_stack_finish_tail": (_i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "tspws_hip_stack": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "tspws_hip_profile_begin": (_i, [_vp, _sz]),
    "tspws_hip_profile_read": (_i, [_vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "tspws_hip_profile_end": (_i, [_vp, C.POINTER(_d), C.POINTER(_sz)]),
    "tspws_hip_stream_launches": (_i, [_vp]),
    "tspws_jackknife_plan": (_i, [_vp, _vp, _sz, _u, _u, _u]),
    "tspws_hip_jackknife": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _u, _vp, _vp, _vp, _vp]),
    "tspws_hip_stack_jackknife": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _u, _vp, _vp, _vp, _vp]),
    "tspws_main_release": (None, []),
    "tspws_main_on": (_i, [_i, _vp, _vp, _vp]),
    "tspws_main_cached_devices": (C.c_ulonglong, []),
    "tspws_hip_finish_shard": (_i, [_vp, _vp, _sz, _u, _u, C.POINTER(_u), C.POINTER(_u)]),
    "tspws_hip_stack_finish_scales": (_i, [_vp, _vp, _sz, _u, _u, _vp, _vp]),
    "tspws_hip_jackknife_buffer": (_i, [_vp, _vp, _u, C.POINTER(_vp), C.POINTER(_sz)]),
    "tspws_hip_jackknife_local": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _u, _vp]),
    "tspws_hip_jackknife_finish": (_i, [_vp, _vp, _sz, _vp, _u, _u, _u, _vp, _vp, _vp, _vp]),
    "tspws_subsampling_plan": (_i, [_vp, _sz, _sz]),
    "tspws_hip_subsample": (_i, [_vp, _vp, _vp, _sz, _sz, _u, _vp, _vp, _vp]),
    "tspws_hip_subsample_sel": (_i, [_vp, _vp, _vp, _sz, _sz, _u, _vp, _vp, _vp, _vp]),
    "tspws_hip_convergence": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tspws_hip_synth": (_i, [_vp, _sz, _sz, _sz, C.c_uint64, _sz, _vp]),
    # several devices of one process (csrc/comm.hip)
    "tspws_hip_comm_create": (_i, [C.POINTER(_vp), _i, _vp]),
    "tspws_hip_comm_destroy": (None, [_vp]),
    "tspws_hip_comm_size": (_i, [_vp]),
    "tspws_hip_comm_device": (_i, [_vp, _i]),
    "tspws_hip_comm_stream": (_vp, [_vp, _i]),
    "tspws_hip_comm_backend": (C.c_char_p, [_vp]),
    "tspws_hip_allreduce_f64": (_i, [_vp, _vp, _sz, _vp]),
    "tspws_hip_reduce_f64": (_i, [_vp, _vp, _sz, _i, _vp]),
    "tspws_shard_range": (None, [_sz, _u, _u, C.POINTER(_sz), C.POINTER(_sz)]),
    "tspws_hip_multi_create": (_i, [C.POINTER(_vp), _i, _vp, _i, _u, _u, _u, _d, _d, _d, _i]),
    "tspws_hip_multi_destroy": (None, [_vp]),
    "tspws_hip_multi_comm": (_vp, [_vp]),
    "tspws_hip_multi_plan": (_vp, [_vp, _i]),
    "tspws_hip_multi_upload": (_i, [_vp, _vp, _sz, _sz, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "tspws_hip_multi_prologue": (_i, [_vp, _vp, _sz, _sz, _sz, _i, _i]),
    "tspws_hip_multi_stack": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _vp]),
    "tspws_hip_multi_stack_jackknife": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _u, _vp, _vp, _vp]),
}


def load():
    """Load libtspws_hip.so and type its entry points.  Raises when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TspwsError(f"{LIB_PATH} is missing: run __graft_entry__.build() / make -C ts-pws_amd")
        # torch wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7) but request it by file
        # name; importing torch FIRST makes the loader hand that same runtime to this library, so a
        # process never ends up with two HIP runtimes (streams / events would not be shared).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the header promises a symbol the library lacks
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc:
        raise TspwsError(f"{what} failed with code {rc}: {load().tspws_hip_last_error().decode()}")


def resolve(params, nsamp, dt=1.0):
    """Resolved copy of a t_tsPWS (tspws_resolve_params; reference ts_pws1f_lib.c:91-124)."""
    p = t_tsPWS.from_buffer_copy(params)
    load().tspws_resolve_params(C.byref(p), nsamp, dt)
    return p


def shard_range(mtr_global, rank, world):
    """Contiguous trace shard [first, first+count) of `rank` (the library's tspws_shard_range)."""
    first = rank * mtr_global // world
    last = (rank + 1) * mtr_global // world
    return first, last - first


class Plan:
    """Device-resident frame (taps, tables, scratch) for one resolved parameter set."""

    def __init__(self, params, N, device=0):
        self.lib = load()
        self.params = t_tsPWS.from_buffer_copy(params)
        h = C.c_void_p()
        p = self.params
        check(self.lib.tspws_hip_plan_create(C.byref(h), p.type, p.J, p.V, N, p.s0, p.b0, p.w0, int(p.uni), device), "plan_create")
        self.h = h
        info = FrameInfo()
        check(self.lib.tspws_hip_plan_info(self.h, C.byref(info)), "plan_info")
        self.info = info
        self.N, self.S, self.ncoef, self.ntaps, self.Cpsi = N, info.S, info.ncoef, info.ntaps, info.Cpsi
        self.device = device

    def tables(self):
        import numpy as np
        S = self.S
        t = dict(scale=np.zeros(S), L=np.zeros(S, np.uint32), c=np.zeros(S, np.int32), cd=np.zeros(S, np.int32),
                 D=np.zeros(S, np.uint32), Ns=np.zeros(S, np.uint32))
        check(self.lib.tspws_hip_plan_tables(self.h, *[t[k].ctypes.data for k in ("scale", "L", "c", "cd", "D", "Ns")]), "plan_tables")
        return t

    def taps(self):
        import numpy as np
        w = np.zeros(self.ntaps, np.complex128)
        wd = np.zeros(self.ntaps, np.complex128)
        check(self.lib.tspws_hip_plan_taps(self.h, w.ctypes.data, wd.ctypes.data), "plan_taps")
        return w, wd

    # ---- device-resident path on torch tensors -------------------------------------
    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def reduce_buffer(self, mtr_global):
        """torch view (float64) of the buffer a multi-GPU caller all-reduces between the halves.

        The view aliases library scratch: it is the buffer of THIS (two-stage / single-stage, Kmax, mtr_global) shape.
        A later call that needs a larger buffer allocates a new block (the old one stays alive until the plan is
        destroyed, so an in-flight collective on it is safe) -- fetch the view again after changing the shape."""
        import torch
        ptr, n = C.c_void_p(), C.c_size_t()
        check(self.lib.tspws_hip_reduce_buffer(self.h, C.byref(self.params), mtr_global, C.byref(ptr), C.byref(n)), "reduce_buffer")
        return _as_tensor(ptr.value, n.value, torch.float64, self.device)

    def _traces(self, traces):
        """(rows, row stride) of a float32 [mtr][N] device tensor after checking what the C ABI cannot see."""
        import torch
        if traces.dtype != torch.float32 or traces.dim() != 2 or traces.shape[1] != self.N:
            raise TspwsError(f"traces must be float32 [mtr][{self.N}], got {traces.dtype} {tuple(traces.shape)}")
        if traces.shape[0] and traces.stride(1) != 1:
            raise TspwsError("traces must be contiguous along the samples (stride(1) == 1)")
        if not traces.is_cuda or (traces.device.index or 0) != self.device:
            raise TspwsError(f"traces live on {traces.device}, the plan on cuda:{self.device}")
        mtr = traces.shape[0]
        ld = traces.stride(0) if mtr > 1 else traces.shape[1]
        if ld < self.N:
            raise TspwsError("overlapping trace rows (stride(0) < N)")
        return mtr, ld

    def _out(self, t, name):
        import torch
        if t.dtype != torch.float32 or t.numel() < self.N or not t.is_contiguous() or not t.is_cuda or (t.device.index or 0) != self.device:
            raise TspwsError(f"{name} must be a contiguous float32 tensor of >= {self.N} samples on cuda:{self.device}")
        return t.data_ptr()

    def stack_local(self, traces, first=0, mtr_global=None):
        mtr, ld = self._traces(traces)
        mtr_global = mtr if mtr_global is None else mtr_global
        check(self.lib.tspws_hip_stack_local(self.h, C.byref(self.params), traces.data_ptr(), ld, mtr, first, mtr_global, self._stream()),
              "stack_local")

    def partial_stacks_range(self, traces, first, mtr_global, g_begin, g_end):
        """Two-stage only: stream the groups [g_begin, g_end) of this shard into rows of the reduce buffer."""
        mtr, ld = self._traces(traces)
        buf = self.reduce_buffer(mtr_global)
        check(self.lib.tspws_hip_partial_stacks_range(self.h, traces.data_ptr(), ld, mtr, first, mtr_global, self.params.Kmax, g_begin, g_end,
                                                      buf.data_ptr(), self.N, self._stream()), "partial_stacks_range")

    def stack_finish(self, mtr_global, ls, ts):
        check(self.lib.tspws_hip_stack_finish(self.h, C.byref(self.params), mtr_global, self._out(ls, "ls"), self._out(ts, "ts"), self._stream()),
              "stack_finish")

    def stack_finish_range(self, mtr_global, g_begin, g_end):
        """Two-stage: transform the reduced partial stacks [g_begin, g_end) and add them to the linear / phase stacks."""
        check(self.lib.tspws_hip_stack_finish_range(self.h, C.byref(self.params), mtr_global, g_begin, g_end, self._stream()),
              "stack_finish_range")

    def stack_finish_tail(self, mtr_global, ls, ts):
        """Weight, inverse transforms, epilogue (after every group went through stack_finish_range)."""
        check(self.lib.tspws_hip_stack_finish_tail(self.h, C.byref(self.params), mtr_global, self._out(ls, "ls"), self._out(ts, "ts"),
                                                   self._stream()), "stack_finish_tail")

    # ---- trace-sharded jackknife (see jackknife_sharded) ------------------------------
    @staticmethod
    def _sel(sel, C_, mtr_global):
        import numpy as np
        sel = np.ascontiguousarray(sel, dtype=np.int8)
        if sel.shape != (C_, mtr_global):
            raise TspwsError(f"selection must be [{C_}][{mtr_global}] (replica x trace of the WHOLE ensemble), got {sel.shape}")
        return sel

    def jackknife_buffer(self, C_):
        """torch view (float64, [C * Kmax * N]) of the replicas' partial-stack rows a multi-GPU caller reduces.  Like
        reduce_buffer the view aliases library scratch of THIS (C, Kmax) shape: an outgrown block stays alive until the plan is
        destroyed (an in-flight collective on it is safe), but later calls use the new block -- fetch the view again."""
        import torch
        ptr, n = C.c_void_p(), C.c_size_t()
        check(self.lib.tspws_hip_jackknife_buffer(self.h, C.byref(self.params), C_, C.byref(ptr), C.byref(n)), "jackknife_buffer")
        return _as_tensor(ptr.value, n.value, torch.float64, self.device)

    def jackknife_local(self, traces, first, mtr_global, sel):
        """One pass over the shard: its sums for the plain groups (reduce_buffer) and for every replica (jackknife_buffer)."""
        mtr, ld = self._traces(traces)
        sel = self._sel(sel, sel.shape[0], mtr_global)
        check(self.lib.tspws_hip_jackknife_local(self.h, C.byref(self.params), traces.data_ptr(), ld, mtr, first, mtr_global, sel.ctypes.data,
                                                 sel.shape[0], self._stream()), "jackknife_local")

    def jackknife_finish(self, mtr_global, sel, c_begin, c_end, ls_out, ts_out, mtr_out):
        """Replicas [c_begin, c_end) from the reduced rows into rows of the [C][N] float32 tensors; sizes into mtr_out (uint32 numpy)."""
        import torch
        import numpy as np
        sel = self._sel(sel, sel.shape[0], mtr_global)
        if not isinstance(mtr_out, np.ndarray) or mtr_out.dtype != np.uint32 or mtr_out.shape != (sel.shape[0],) or not mtr_out.flags.c_contiguous:
            raise TspwsError(f"mtr_out must be a contiguous uint32 numpy array of {sel.shape[0]} entries")
        for t, name in ((ls_out, "ls_out"), (ts_out, "ts_out")):
            if t.dtype != torch.float32 or tuple(t.shape) != (sel.shape[0], self.N) or not t.is_contiguous() or not t.is_cuda or \
                    (t.device.index or 0) != self.device:
                raise TspwsError(f"{name} must be a contiguous float32 [{sel.shape[0]}][{self.N}] tensor on cuda:{self.device}")
        check(self.lib.tspws_hip_jackknife_finish(self.h, C.byref(self.params), mtr_global, sel.ctypes.data, sel.shape[0], c_begin, c_end,
                                                  ls_out.data_ptr(), ts_out.data_ptr(), mtr_out.ctypes.data, self._stream()), "jackknife_finish")

    # ---- scale-sharded finish stage (see stack_sharded) --------------------------------
    def finish_shard(self, mtr_global, rank, world):
        """Scales [s_begin, s_end) that `rank` of `world` finishes, or None when this plan / parameter set has no sharded finish."""
        a, b = C.c_uint(), C.c_uint()
        rc = self.lib.tspws_hip_finish_shard(self.h, C.byref(self.params), mtr_global, rank, world, C.byref(a), C.byref(b))
        if rc == 1:
            return None
        check(rc, "finish_shard")
        return a.value, b.value

    def stack_finish_scales(self, mtr_global, s_begin, s_end, x2):
        """This rank's share of the finish stage: x2 (float64 [2 N], cuda) receives the partial reconstructions of the scales."""
        import torch
        if x2.dtype != torch.float64 or x2.numel() != 2 * self.N or not x2.is_contiguous() or not x2.is_cuda or (x2.device.index or 0) != self.device:
            raise TspwsError(f"x2 must be a contiguous float64 tensor of {2 * self.N} values on cuda:{self.device}")
        check(self.lib.tspws_hip_stack_finish_scales(self.h, C.byref(self.params), mtr_global, s_begin, s_end, x2.data_ptr(), self._stream()),
              "stack_finish_scales")

    def epilogue(self, x2, mtr_global, ls, ts):
        """ls = (float)x2[N:] / mtr, ts = (float)x2[:N]  (reference epilogue, ts_pws1f_lib.c:233-241)."""
        import torch
        if x2.dtype != torch.float64 or x2.numel() != 2 * self.N or not x2.is_contiguous() or not x2.is_cuda or (x2.device.index or 0) != self.device:
            raise TspwsError(f"x2 must be a contiguous float64 tensor of {2 * self.N} values on cuda:{self.device}")
        check(self.lib.tspws_hip_epilogue(self._out(ls, "ls"), self._out(ts, "ts"), x2.data_ptr() + 8 * self.N, x2.data_ptr(), self.N, mtr_global,
                                          self._stream()), "epilogue")

    def stack(self, traces, first=0, mtr_global=None, group=None):
        """ls, tsPWS (float32 cuda tensors) of a shard of HBM-resident traces; see stack_sharded."""
        return stack_sharded(self, traces, first, mtr_global, group)

    def stack_jackknife(self, traces, sel, ls=None, ts=None):
        """Two-stage stack AND its jackknife replicas from ONE pass over the traces (tspws_hip_stack_jackknife).
        `sel` = [C][mtr] int8 selection (tspws_jackknife_plan).  Returns ls, ts, ls_out[C][N], ts_out[C][N], mtr_out[C]."""
        import numpy as np
        import torch
        mtr, ld = self._traces(traces)
        sel = self._sel(sel, sel.shape[0], mtr)
        Cn = sel.shape[0]
        ls = torch.empty(self.N, dtype=torch.float32, device=traces.device) if ls is None else ls
        ts = torch.empty(self.N, dtype=torch.float32, device=traces.device) if ts is None else ts
        ls_out = torch.empty((Cn, self.N), dtype=torch.float32, device=traces.device)
        ts_out = torch.empty((Cn, self.N), dtype=torch.float32, device=traces.device)
        mtr_out = np.zeros(Cn, np.uint32)
        check(self.lib.tspws_hip_stack_jackknife(self.h, C.byref(self.params), traces.data_ptr(), ld, mtr, self._out(ls, "ls"), self._out(ts, "ts"),
                                                 sel.ctypes.data, Cn, ls_out.data_ptr(), ts_out.data_ptr(), mtr_out.ctypes.data, self._stream()),
              "stack_jackknife")
        return ls, ts, ls_out, ts_out, mtr_out

    def profile_begin(self, max_calls):
        """Record HIP events inside the next `max_calls` stack_single calls (start, end of the streaming stage, end)."""
        check(self.lib.tspws_hip_profile_begin(self.h, max_calls), "profile_begin")

    def profile_read(self):
        """(stage_ms[], call_ms[]) of the recorded calls (numpy float64); synchronises the device."""
        import numpy as np
        n = C.c_size_t()
        check(self.lib.tspws_hip_profile_read(self.h, None, None, 0, C.byref(n)), "profile_read")
        st, ca = np.zeros(n.value), np.zeros(n.value)
        check(self.lib.tspws_hip_profile_read(self.h, st.ctypes.data, ca.ctypes.data, n.value, C.byref(n)), "profile_read")
        return st, ca

    def stack_single(self, traces, ls=None, ts=None):
        """Whole call on ONE GPU through tspws_hip_stack (stack_local + stack_finish in one C call)."""
        import torch
        mtr, ld = self._traces(traces)
        ls = torch.empty(self.N, dtype=torch.float32, device=traces.device) if ls is None else ls
        ts = torch.empty(self.N, dtype=torch.float32, device=traces.device) if ts is None else ts
        check(self.lib.tspws_hip_stack(self.h, C.byref(self.params), traces.data_ptr(), ld, mtr, self._out(ls, "ls"), self._out(ts, "ts"),
                                       self._stream()), "stack")
        return ls, ts

    def close(self):
        if getattr(self, "h", None):
            self.lib.tspws_hip_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SCHEDULES = ("single", "split", "sharded-finish")


def stack_sharded(plan, traces, first=0, mtr_global=None, group=None, schedule=None):
    """One tspws_main-equivalent call over a trace-sharded ensemble.

    Every rank holds a contiguous shard `traces` whose first row is global trace `first` of
    `mtr_global`.  The shard-local half produces a sum over traces (two-stage: the Kmax partial
    stacks, group index taken from the GLOBAL trace index; single-stage: ST||PS); ONE all-reduce
    (RCCL over xGMI on GPUs, backend "nccl") adds the shards; every rank then finishes (K forward
    CWTs, weight, two inverses) redundantly -- that part is tiny.  With world_size 1 (or no process
    group) there is no collective at all.  `plan` only needs stack_local / reduce_buffer /
    stack_finish / N, so the CPU tests drive the same orchestration with an oracle-backed plan.

    `schedule` (N > 1, two-stage; default: $TSPWS_SCHEDULE or "single") -- three ways to place the one logical
    reduction of P[Kmax][N], selectable so that a multi-GPU node can A/B them (bench.py --schedule):
      "single"          north_star's wording: local half, ONE all-reduce of the whole buffer, redundant finish on every rank
                        (bit-identical to the one-GPU result when the shards add exactly);
      "split"           the groups in two pieces K/2 | K/2: the first reduction runs while the second piece is streamed, the
                        second while the first half of the groups is transformed (redundant finish in pieces);
      "sharded-finish"  pieces K-2 | 2, then every rank finishes only its share of the scales and a 2 N-double all-reduce adds
                        the partial reconstructions (falls back to "split" when the plan has no sharded finish)."""
    import torch
    import torch.distributed as dist
    mtr_global = traces.shape[0] if mtr_global is None else mtr_global
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    K = getattr(getattr(plan, "params", None), "Kmax", 0)
    schedule = schedule or os.environ.get("TSPWS_SCHEDULE") or "single"
    if schedule not in SCHEDULES:
        raise TspwsError(f"schedule must be one of {SCHEDULES}, got {schedule!r}")
    if distributed and schedule != "single" and callable(getattr(plan, "partial_stacks_range", None)) and K >= 2 and K <= mtr_global:
        # two-stage: the sum is row-separable, so the all-reduce of the first half of the groups runs (on the collective's
        # own stream) while the second half is still being streamed -- still one logical reduction of P[Kmax][N]
        shard = _finish_shard(plan, mtr_global, group) if schedule == "sharded-finish" else None
        half = split_groups(K, shard is not None)
        buf = plan.reduce_buffer(mtr_global).view(K, plan.N)
        plan.partial_stacks_range(traces, first, mtr_global, 0, half)
        w1 = dist.all_reduce(buf[:half], op=dist.ReduceOp.SUM, group=group, async_op=True)
        plan.partial_stacks_range(traces, first, mtr_global, half, K)
        w2 = dist.all_reduce(buf[half:], op=dist.ReduceOp.SUM, group=group, async_op=True)
        ls = torch.empty(plan.N, dtype=torch.float32, device=traces.device)
        ts = torch.empty(plan.N, dtype=torch.float32, device=traces.device)
        if shard is not None:
            # scale-sharded finish: every rank transforms / weights / reconstructs only its share of the scales of the
            # reduced partial stacks; the partial reconstructions (2 N doubles) are added and every rank ends with the outputs
            w1.wait()
            w2.wait()
            x2 = torch.empty(2 * plan.N, dtype=torch.float64, device=traces.device)
            plan.stack_finish_scales(mtr_global, shard[0], shard[1], x2)
            dist.all_reduce(x2, op=dist.ReduceOp.SUM, group=group)
            plan.epilogue(x2, mtr_global, ls, ts)
            return ls, ts
        w1.wait()
        if callable(getattr(plan, "stack_finish_range", None)):
            # the transforms of the first half of the groups run while the second half is still being reduced
            plan.stack_finish_range(mtr_global, 0, half)
            w2.wait()
            plan.stack_finish_range(mtr_global, half, K)
            plan.stack_finish_tail(mtr_global, ls, ts)
        else:
            w2.wait()
            plan.stack_finish(mtr_global, ls, ts)
        return ls, ts
    plan.stack_local(traces, first, mtr_global)
    if distributed:
        dist.all_reduce(plan.reduce_buffer(mtr_global), op=dist.ReduceOp.SUM, group=group)
    ls = torch.empty(plan.N, dtype=torch.float32, device=traces.device)
    ts = torch.empty(plan.N, dtype=torch.float32, device=traces.device)
    plan.stack_finish(mtr_global, ls, ts)
    return ls, ts


def jackknife_sharded(plan, traces, sel, first=0, mtr_global=None, group=None):
    """Two-stage stack AND its jackknife replicas over a trace-sharded ensemble ("Jackknife sharding").

    `sel` is the [C][mtr_global] selection of the whole ensemble (tspws_jackknife_plan), identical on every rank.  Each rank
    walks its shard ONCE (plan.jackknife_local): the sums of its traces for the Kmax plain groups and for the Kmax groups of
    every replica -- group indices come from the GLOBAL trace order, so the shards' rows simply add.  The plain rows are
    all-reduced and every rank finishes the stack (as in stack_sharded).  The replicas are SHARDED for the finish stage,
    which is where their time goes (Kmax transforms + an inverse each): rank r owns the contiguous block of replicas
    [r C / world, (r + 1) C / world), their rows are reduced to that rank only (one reduction per owner instead of one
    all-reduce of all C x Kmax rows on every rank), the owner finishes its block in one batched call, and one all-reduce of
    the [C][N] float outputs (zeros from the non-owners: exact) hands every rank all replicas.  Returns ls, ts, ls_out[C][N], ts_out[C][N], mtr_out[C]."""
    import numpy as np
    import torch
    import torch.distributed as dist
    mtr_global = traces.shape[0] if mtr_global is None else mtr_global
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    sel = np.ascontiguousarray(sel, dtype=np.int8)
    Cn = sel.shape[0]
    K, N = plan.params.Kmax, plan.N
    plan.jackknife_local(traces, first, mtr_global, sel)
    dev = traces.device
    ls = torch.empty(N, dtype=torch.float32, device=dev)
    ts = torch.empty(N, dtype=torch.float32, device=dev)
    ls_out = torch.zeros((Cn, N), dtype=torch.float32, device=dev)
    ts_out = torch.zeros((Cn, N), dtype=torch.float32, device=dev)
    mtr_out = np.zeros(Cn, np.uint32)
    # replica c belongs to the rank whose CONTIGUOUS block [r * C // world, (r + 1) * C // world) holds it: one reduction per
    # owner covers all its rows, and the owner finishes its block in one batched call (forward launch, inverses in pairs)
    def block(r):
        return r * Cn // world, (r + 1) * Cn // world
    works = []
    if distributed:
        rows = plan.jackknife_buffer(Cn).view(Cn, K * N)
        works.append(dist.all_reduce(plan.reduce_buffer(mtr_global), op=dist.ReduceOp.SUM, group=group, async_op=True))
        for r in range(world):
            c0, c1 = block(r)
            if c1 > c0:
                works.append(dist.reduce(rows[c0:c1], dst=dist.get_global_rank(group, r) if group is not None else r, op=dist.ReduceOp.SUM,
                                         group=group, async_op=True))
    for w in works[:1]:
        w.wait()
    plan.stack_finish(mtr_global, ls, ts)  # runs while the replicas' rows are still being reduced
    for w in works[1:]:
        w.wait()
    c0, c1 = block(rank)
    if c1 > c0:
        plan.jackknife_finish(mtr_global, sel, c0, c1, ls_out, ts_out, mtr_out)
    if distributed:
        dist.all_reduce(ls_out, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(ts_out, op=dist.ReduceOp.SUM, group=group)
        cnt = torch.from_numpy(mtr_out.astype(np.int64)).to(dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
        mtr_out = cnt.cpu().numpy().astype(np.uint32)
    return ls, ts, ls_out, ts_out, mtr_out


def _finish_shard(plan, mtr_global, group=None):
    """This rank's share of the scales for the sharded finish stage, or None (plan without one, world 1, TSPWS_SHARD_FINISH=0).
    The decision is AGREED across the ranks: each one looks at its own environment and plan, so a rank that would not shard
    (a different TSPWS_SHARD_FINISH, a check-kernel switch) would otherwise issue a different sequence of collectives and
    hang the job -- one MIN all-reduce of a flag makes every rank take the redundant finish unless all of them can shard."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world < 2:
        return None
    # agreed once per (group, ensemble size, frame): no collective in later calls.  The cache is module-level and keyed by the
    # frame's parameters, not by the plan object: a rank that rebuilds its plan hits the same entry as the ranks that kept
    # theirs, so the ranks cannot diverge into different collective sequences.
    pr = getattr(plan, "params", None)
    frame = (pr.type, pr.J, pr.V, pr.s0, pr.b0, pr.w0, int(pr.uni), pr.Kmax) if pr is not None else (id(plan),)
    # keyed on the process group AND on torch's count of groups created so far (it grows with every init_process_group / new_group, on
    # every rank alike): after destroy + init the id of the old group object may be reused, but the count has moved on, so all ranks miss
    # together and repeat the agreement collective -- no rank skips a collective that a peer still issues.  (No reference to the group is
    # kept: a process group held by a module-level table is torn down at interpreter exit with its threads still running.)
    try:
        gen = dist.distributed_c10d._world.group_count
    except Exception:
        gen = None
    key = (id(group) if group is not None else None, gen, world, dist.get_rank(group), mtr_global, getattr(plan, "N", 0), frame,
           os.environ.get("TSPWS_SHARD_FINISH", "1"))
    if gen is not None and key in _SHARD_AGREED:
        return _SHARD_AGREED[key]
    mine = None
    if os.environ.get("TSPWS_SHARD_FINISH", "1") != "0" and callable(getattr(plan, "finish_shard", None)):
        mine = plan.finish_shard(mtr_global, dist.get_rank(group), world)
    # the flag lives on the PLAN's device (not torch's current device: a caller that never ran torch.cuda.set_device would put
    # every rank's tensor on cuda:0 and RCCL would hang)
    dev = f"cuda:{getattr(plan, 'device', 0)}" if dist.get_backend(group) == "nccl" else "cpu"
    flag = torch.tensor([1 if mine is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    agreed = mine if int(flag.item()) == 1 else None
    _SHARD_AGREED[key] = agreed
    return agreed


_SHARD_AGREED = {}


def split_groups(K, sharded_finish=False):
    """First piece of the two-piece streaming / reduction schedule.  Redundant finish: about half the groups -- the second
    reduction then hides behind the transforms of the first half.  Scale-sharded finish (nothing to hide behind): all but
    the last two groups, so that only a small reduction is left exposed.  EVEN when possible -- the streaming pass
    launches the groups two at a time (one workgroup per CU at N = 131072), so an odd piece would end on a launch that
    fills half the CUs."""
    half = K - 2 if (sharded_finish and K > 3) else K // 2
    if half >= 2 and half % 2:
        half -= 1
    return half


def _as_tensor(ptr, count, dtype, device):
    """Zero-copy torch view of library-owned device memory (via __cuda_array_interface__)."""
    import torch
    typestr = {torch.float64: "<f8", torch.float32: "<f4"}[dtype]

    class _Holder:
        __cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}

    return torch.as_tensor(_Holder(), device=f"cuda:{device}")


def synth(mtr, N, seed=0, first=0, device=0, pad=0):
    """Seeded synthetic ensemble generated on the device: float32 [mtr][N] view of rows `N + pad` samples apart (pad > 0:
    the trace rows do not all start at the same offset of the memory interleave)."""
    import torch
    ld = N + pad
    buf = torch.empty((mtr, ld), dtype=torch.float32, device=f"cuda:{device}")
    check(load().tspws_hip_synth(buf.data_ptr(), mtr, N, ld, seed, first, Plan._stream()), "synth")
    return buf[:, :N] if pad else buf
