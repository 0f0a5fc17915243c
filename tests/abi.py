"""ctypes view of the drop-in boundary (include/ts_pws1f_lib.h) and loaders for
the three libraries that implement it:

  * hip()     -- the product: ts-pws_amd/lib/libtspws_hip.so (HIP kernels + C host)
  * oracle()  -- oracle/liboracle.so, this repo's CPU restatement (checker only)
  * ref()     -- oracle/_ref/libtspws_ref.so, the reference compiled in place
                 (only where it has been built; never required on the GPU box)

Struct layouts follow /root/reference/src/ts_pws1f_lib.h:19-102.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

time_t = C.c_long


class t_tsPWS(C.Structure):
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


class t_hdr(C.Structure):
    _fields_ = [
        ("max", C.c_int), ("mtr", C.c_uint),
        ("evla", C.c_float), ("evlo", C.c_float), ("stla", C.c_float), ("stlo", C.c_float),
        ("stel", C.c_float), ("dt", C.c_float), ("beg", C.c_float),
        ("net1", C.c_char * 9), ("sta1", C.c_char * 9), ("loc1", C.c_char * 9), ("chn1", C.c_char * 9),
        ("net2", C.c_char * 9), ("sta2", C.c_char * 9), ("loc2", C.c_char * 9), ("chn2", C.c_char * 9),
    ]


class t_tsPWS_out(C.Structure):
    _fields_ = [
        ("ls", C.POINTER(C.c_float)), ("tsPWS", C.POINTER(C.c_float)),
        ("ls_sim", C.POINTER(C.c_double)), ("tsPWS_sim", C.POINTER(C.c_double)),
        ("ls_misfit", C.POINTER(C.c_double)), ("tsPWS_misfit", C.POINTER(C.c_double)),
        ("ls_steps", C.POINTER(C.c_float)), ("tsPWS_steps", C.POINTER(C.c_float)),
        ("ls_subsmpl", C.POINTER(C.POINTER(C.c_float))), ("tsPWS_subsmpl", C.POINTER(C.POINTER(C.c_float))),
        ("mtr_subsmpl", C.POINTER(C.c_uint)),
        ("M", C.c_uint), ("N", C.c_uint), ("mtr", C.c_uint),
    ]


class t_data(C.Structure):
    _fields_ = [
        ("sigall", C.POINTER(C.c_float)), ("time", C.POINTER(time_t)), ("lag0", C.POINTER(C.c_float)),
        ("hdr", t_hdr), ("reference", C.POINTER(C.c_float)),
    ]


assert C.sizeof(t_tsPWS) == 184 and C.sizeof(t_hdr) == 108
assert C.sizeof(t_tsPWS_out) == 104 and C.sizeof(t_data) == 144

W0_DEFAULT = math.pi * math.sqrt(2 / math.log(2))


def default_params(**kw):
    """CLI defaults, /root/reference/src/ts_pws1f.c:140-142."""
    p = t_tsPWS()
    p.type, p.uni, p.J, p.V = -1, 0, 0, 4
    p.s0, p.b0, p.w0, p.wu = 2.0, 1.0, W0_DEFAULT, 2.0
    p.fmin, p.Q, p.cycle = 0.0, 0.0, 2.0
    for k, v in kw.items():
        if k == "V":
            p.lVfix = 1
        if k == "s0":
            p.ls0fix = 1
        if k == "b0":
            p.lb0fix = 1
        if k == "w0" and "w0set" not in kw:
            p.w0set = 4
        setattr(p, k, v)
    return p


def binomial(n, d):
    return math.comb(n, d)


def srand(seed=1):
    """Reset libc's rand(): the random-subsampling masks of the reference (and of this engine's host
    side) come from rand(), so runs are comparable only from the same generator state."""
    C.CDLL(None).srand(seed)


def run_main(lib_fn, params, traces, dt=1.0, beg=0.0, times=None, reference=None):
    """Call a tspws_main-shaped entry point on a copy of `traces` (float32 [mtr][max]).
    Returns dict(rc, ls, tsPWS, sigall, params, [jk_ls, jk_ts, jk_mtr | sub_ls, sub_ts], [conv_*])."""
    x = np.ascontiguousarray(traces, dtype=np.float32).copy()
    mtr, mx = x.shape
    p = t_tsPWS.from_buffer_copy(params)
    out = t_tsPWS_out()
    ls = np.zeros(mx, np.float32)
    ts = np.zeros(mx, np.float32)
    fp = C.POINTER(C.c_float)
    out.ls = ls.ctypes.data_as(fp)
    out.tsPWS = ts.ctypes.data_as(fp)
    out.N, out.mtr = mx, (p.Nmax or mtr)
    keep = []
    res = {}
    M = 0
    if p.jackknife_n and 0 < p.jackknife_d < p.jackknife_n:
        M, key = binomial(p.jackknife_n, p.jackknife_d), "jk"
    elif p.subsmpl_N and 0 <= p.subsmpl_p <= 1:
        M, key = p.subsmpl_N, "sub"
    if M:
        out.M = M
        jl = np.zeros((M, mx), np.float32)
        jt = np.zeros((M, mx), np.float32)
        jm = np.zeros(M, np.uint32)
        rows_l = (fp * M)(*[jl[i].ctypes.data_as(fp) for i in range(M)])
        rows_t = (fp * M)(*[jt[i].ctypes.data_as(fp) for i in range(M)])
        out.ls_subsmpl = C.cast(rows_l, C.POINTER(fp))
        out.tsPWS_subsmpl = C.cast(rows_t, C.POINTER(fp))
        out.mtr_subsmpl = jm.ctypes.data_as(C.POINTER(C.c_uint))
        keep += [rows_l, rows_t]
        res.update({key + "_ls": jl, key + "_ts": jt, key + "_mtr": jm})
    if p.convergence:
        n = out.mtr
        conv = {k: np.zeros(n) for k in ("ls_sim", "tsPWS_sim", "ls_misfit", "tsPWS_misfit")}
        dp = C.POINTER(C.c_double)
        out.ls_sim, out.tsPWS_sim = conv["ls_sim"].ctypes.data_as(dp), conv["tsPWS_sim"].ctypes.data_as(dp)
        out.ls_misfit, out.tsPWS_misfit = conv["ls_misfit"].ctypes.data_as(dp), conv["tsPWS_misfit"].ctypes.data_as(dp)
        res.update({"conv_" + k: v for k, v in conv.items()})
        if p.AllSteps:
            sl = np.zeros((n, mx), np.float32)
            st = np.zeros((n, mx), np.float32)
            out.ls_steps, out.tsPWS_steps = sl.ctypes.data_as(fp), st.ctypes.data_as(fp)
            res.update(conv_ls_steps=sl, conv_ts_steps=st)
    d = t_data()
    d.sigall = x.ctypes.data_as(fp)
    if times is not None:
        tarr = np.ascontiguousarray(times, dtype=np.int64)
        d.time = tarr.ctypes.data_as(C.POINTER(time_t))
        keep.append(tarr)
    if reference is not None:
        rarr = np.ascontiguousarray(reference, dtype=np.float32)
        d.reference = rarr.ctypes.data_as(fp)
        keep.append(rarr)
    d.hdr.max, d.hdr.mtr, d.hdr.dt, d.hdr.beg = mx, mtr, dt, beg
    rc = lib_fn(C.byref(p), C.byref(out), C.byref(d))
    res.update(rc=rc, ls=ls, tsPWS=ts, sigall=x, params=p)
    return res


# ---------------------------------------------------------------- libraries --
_cache = {}


def _load(path):
    if path not in _cache:
        _cache[path] = C.CDLL(path)
    return _cache[path]


def oracle():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])
    lib = _load(path)
    if not getattr(lib, "_typed", False):
        vp, u, d, sz, i = C.c_void_p, C.c_uint, C.c_double, C.c_size_t, C.c_int
        lib.orc_frame_create.restype = vp
        lib.orc_frame_create.argtypes = [i, u, u, u, d, d, d, i]
        lib.orc_frame_destroy.argtypes = [vp]
        lib.orc_frame_S.restype = u
        lib.orc_frame_S.argtypes = [vp]
        lib.orc_frame_ncoef.restype = sz
        lib.orc_frame_ncoef.argtypes = [vp]
        lib.orc_frame_ntaps.restype = sz
        lib.orc_frame_ntaps.argtypes = [vp]
        lib.orc_frame_cpsi.restype = d
        lib.orc_frame_cpsi.argtypes = [vp]
        lib.orc_frame_tables.argtypes = [vp] + [vp] * 6
        lib.orc_frame_taps.argtypes = [vp, vp, vp]
        lib.orc_forward.argtypes = [vp, vp, vp]
        lib.orc_inverse.argtypes = [vp, vp, vp]
        lib.orc_accumulate.argtypes = [vp, vp, vp, sz]
        lib.orc_weight.argtypes = [vp, vp, vp, sz, u, u, d, i]
        lib.orc_partial_stacks.argtypes = [vp, vp, sz, sz, u]
        lib.orc_resolve.argtypes = [vp, u, C.c_float]
        lib.orc_subsampling_plan.restype = i
        lib.orc_subsampling_plan.argtypes = [vp, sz, sz]
        lib.orc_jackknife_plan.restype = i
        lib.orc_jackknife_plan.argtypes = [vp, vp, sz, u, u, u]
        lib.orc_tspws_main.restype = i
        lib.orc_tspws_main_mt.restype = i
        lib._typed = True
    return lib


def ref_path():
    return os.path.join(ROOT, "oracle", "_ref", "libtspws_ref.so")


def ref():
    """The reference itself (None where it has not been built)."""
    path = ref_path()
    if not os.path.exists(path):
        return None
    lib = _load(path)
    lib.tspws_main.restype = C.c_int
    return lib


class OracleFrame:
    """numpy-friendly wrapper over the oracle's frame object."""

    def __init__(self, type=-1, J=0, V=4, N=0, s0=2.0, b0=1.0, w0=W0_DEFAULT, uni=0):
        self.lib = oracle()
        self.h = self.lib.orc_frame_create(type, J, V, N, s0, b0, w0, uni)
        assert self.h
        self.N = N
        self.S = self.lib.orc_frame_S(self.h)
        self.ncoef = self.lib.orc_frame_ncoef(self.h)
        self.ntaps = self.lib.orc_frame_ntaps(self.h)
        self.Cpsi = self.lib.orc_frame_cpsi(self.h)
        S = self.S
        self.scale = np.zeros(S)
        self.L = np.zeros(S, np.uint32)
        self.c = np.zeros(S, np.int32)
        self.cd = np.zeros(S, np.int32)
        self.D = np.zeros(S, np.uint32)
        self.Ns = np.zeros(S, np.uint32)
        self.lib.orc_frame_tables(self.h, *[a.ctypes.data for a in (self.scale, self.L, self.c, self.cd, self.D, self.Ns)])

    @classmethod
    def from_params(cls, p, N):
        return cls(p.type, p.J, p.V, N, p.s0, p.b0, p.w0, p.uni)

    def taps(self):
        w = np.zeros(self.ntaps, np.complex128)
        wd = np.zeros(self.ntaps, np.complex128)
        self.lib.orc_frame_taps(self.h, w.ctypes.data, wd.ctypes.data)
        return w, wd

    def forward(self, x):
        x = np.ascontiguousarray(x, np.float64)
        Y = np.zeros(self.ncoef, np.complex128)
        self.lib.orc_forward(self.h, x.ctypes.data, Y.ctypes.data)
        return Y

    def inverse(self, Y):
        Y = np.ascontiguousarray(Y, np.complex128)
        x = np.zeros(self.N)
        self.lib.orc_inverse(self.h, Y.ctypes.data, x.ctypes.data)
        return x

    def __del__(self):
        try:
            self.lib.orc_frame_destroy(self.h)
        except Exception:
            pass


def resolve(params, nsamp, dt=1.0):
    """Return a resolved copy of params (oracle restatement of ts_pws1f_lib.c:91-124)."""
    p = t_tsPWS.from_buffer_copy(params)
    oracle().orc_resolve(C.byref(p), nsamp, dt)
    return p


def relerr(a, b):
    """Parity metric of SURVEY.md 8(d): max|a-b| / max|b|."""
    a = np.asarray(a)
    b = np.asarray(b)
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a, b = a.astype(dt), b.astype(dt)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / den) if den > 0 else float(np.max(np.abs(a - b)))


# ------------------------------------------------------------ synthetic data --
def synth_traces(mtr, N, seed=0, first=0):
    """Seeded synthetic ensemble of SURVEY.md 8(d): coherent wavelet packet +
    iid uniform noise from a counter-based 64-bit mixer (same integer recipe as
    the device generator tspws_hip_synth)."""
    n = np.arange(N, dtype=np.float64)
    T = 200.0
    sig = 0.2 * np.sin(2 * np.pi * (n - N / 2) / T) * np.exp(-0.5 * ((n - N / 2) / (0.05 * N)) ** 2)
    i = (np.arange(first, first + mtr, dtype=np.uint64)[:, None] * np.uint64(N) + np.arange(N, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        z = i + np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24)) - 0.5
    return (sig[None, :] + u).astype(np.float32)


# ------------------------------------------------------------------- SAC files --
def write_sac(path, data, delta, b, year=None, jday=None, kstnm="STA", big_endian=False):
    """Minimal SAC v6 writer for test inputs (632-byte header + float32 samples)."""
    data = np.asarray(data, np.float32)
    fl = np.full(70, -12345.0, np.float32)
    it = np.full(40, -12345, np.int32)
    fl[0], fl[5], fl[6] = delta, b, b + (len(data) - 1) * delta
    it[6], it[9], it[15], it[35], it[36], it[37], it[38], it[39] = 6, len(data), 1, 1, 0, 1, 1, 0
    if year is not None:
        it[0], it[1], it[2], it[3], it[4], it[5] = year, jday, 0, 0, 0, 0
    k = bytearray(b"-12345  " * 24)
    k[8:24] = b"-12345          "
    k[0:8] = kstnm.encode().ljust(8)[:8]
    e = ">" if big_endian else "<"
    with open(path, "wb") as f:
        f.write(fl.astype(e + "f4").tobytes() + it.astype(e + "i4").tobytes() + bytes(k) + data.astype(e + "f4").tobytes())


def read_sac(path):
    raw = open(path, "rb").read()
    fl = np.frombuffer(raw, "<f4", 70, 0)
    it = np.frombuffer(raw, "<i4", 40, 280)
    return dict(f=fl, i=it, k=raw[440:632], data=np.frombuffer(raw, "<f4", int(it[9]), 632).copy())
