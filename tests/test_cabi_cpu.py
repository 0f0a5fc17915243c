"""CPU-side checks of the product library: it loads, exports every symbol the headers in
include/ declare, its host logic matches the goldens/oracle, and it FAILS LOUDLY (no CPU
fallback) when no HIP device is present.  No compute call needs a GPU here."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

import abi

tspws = importlib.import_module("ts-pws_amd")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(tspws.LIB_PATH):
        tspws.build()
    return tspws.load()


def declared_functions(header):
    txt = open(os.path.join(abi.ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(tspws_[a-z0-9_]+)\s*\(", txt))


def test_exports_every_declared_symbol(lib):
    names = declared_functions("tspws_hip.h") | declared_functions("ts_pws1f_lib.h")
    assert "tspws_main" in names and len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
        assert n in tspws.SYMBOLS, f"{n} has no ctypes signature in ts-pws_amd"
    assert set(tspws.SYMBOLS) == names


def test_struct_layout_matches_reference_abi():
    assert C.sizeof(tspws.t_tsPWS) == 184
    assert tspws.t_tsPWS.Kmax.offset == 148 and tspws.t_tsPWS.fileconv.offset == 176
    assert abi.t_tsPWS_out.M.offset == 88 and abi.t_data.hdr.offset == 24


def test_resolve_params_matches_reference(lib, golden):
    g = golden["mains"]
    names = sorted({k.split("/")[0] for k in g.files if "/in/" in k})
    for name in names:
        p = abi.t_tsPWS()
        for k in [k for k in g.files if k.startswith(f"{name}/in/")]:
            setattr(p, k.split("/")[-1], g[k].item())
        tag = str(g[f"{name}/input"])
        n = 2048 if tag == "X" else 2047 if tag == "Xodd" else g[f"{name}/x"].shape[1]
        lib.tspws_resolve_params(C.byref(p), n, float(g[f"{name}/dt"]))
        for k in ("J", "V", "s0", "b0", "w0", "fmin"):
            assert getattr(p, k) == g[f"{name}/out/{k}"].item(), (name, k)


def test_jackknife_plan_matches_oracle(lib):
    orc = abi.oracle()
    rng = np.random.default_rng(5)
    for (n, d, mtr) in [(4, 1, 16), (5, 2, 40), (10, 1, 365), (6, 3, 100), (12, 2, 731)]:
        Cn = abi.binomial(n, d)
        times = (1262304000 + 86400 * rng.integers(0, 1200, mtr)).astype(np.int64)
        a = np.zeros((Cn, mtr), np.int8)
        b = np.zeros((Cn, mtr), np.int8)
        assert lib.tspws_jackknife_plan(a.ctypes.data, times.ctypes.data, mtr, d, n, Cn) == 0
        assert orc.orc_jackknife_plan(b.ctypes.data, times.ctypes.data, mtr, d, n, Cn) == 0
        np.testing.assert_array_equal(a, b)
        assert set(np.unique(a)) <= {0, 1}
    # leap-day quirk: tm_yday == 365 lands in bin n, which is never deleted
    t = np.array([1262304000 + 86400 * (365 * 2 + 365)], np.int64)  # 2012-12-31, yday 365
    s = np.zeros((4, 1), np.int8)
    assert lib.tspws_jackknife_plan(s.ctypes.data, t.ctypes.data, 1, 1, 4, 4) == 0
    assert s.sum() == 4
    z = np.zeros(3, np.int64)
    assert lib.tspws_jackknife_plan(s.ctypes.data, z.ctypes.data, 1, 1, 4, 4) == -2


def test_null_arguments(lib):
    assert lib.tspws_main(None, None, None) == -1


def test_no_cpu_fallback_without_device(lib):
    if lib.tspws_hip_device_count() > 0:
        pytest.skip("a HIP device is present")
    r = abi.run_main(lib.tspws_main, abi.default_params(), abi.synth_traces(4, 256, seed=1))
    assert r["rc"] == 5  # TSPWS_E_NODEV
    assert not r["ls"].any() and not r["tsPWS"].any()
    h = C.c_void_p()
    assert lib.tspws_hip_plan_create(C.byref(h), -1, 4, 4, 256, 2.0, 1.0, abi.W0_DEFAULT, 0, 0) == 5
    assert b"device" in lib.tspws_hip_last_error()


def test_per_device_entry_and_cache_table(lib):
    """tspws_main_on (additive: include/tspws_hip.h) names the device itself; the cache is a table with one slot and one lock per
    device.  Without a GPU the table logic is what can run: out-of-range and absent devices fail with 5 before anything is cached,
    an EMPTY ensemble returns 0 like the reference (ts_pws1f_lib.c:194) with or without a device, NULL arguments give -1, and
    releasing an empty table is harmless."""
    assert lib.tspws_main_on(0, None, None, None) == -1
    X = abi.synth_traces(4, 256, seed=1)
    fn = lambda dev: (lambda p, o, d: lib.tspws_main_on(dev, p, o, d))
    for dev in (-1, 64, 1000):
        assert abi.run_main(fn(dev), abi.default_params(), X)["rc"] == 5
    if lib.tspws_hip_device_count() == 0:
        assert abi.run_main(fn(0), abi.default_params(), X)["rc"] == 5
        assert lib.tspws_main_cached_devices() == 0
    # no traces: the reference returns 0 and leaves the outputs alone -- also where there is no device at all
    r = abi.run_main(fn(0), abi.default_params(), X[:0].reshape(0, 256))
    assert r["rc"] == 0 and not r["ls"].any()
    r = abi.run_main(lib.tspws_main, abi.default_params(Kmax=3), X[:0].reshape(0, 256))
    assert r["rc"] == 0
    lib.tspws_main_release()
    assert lib.tspws_main_cached_devices() == 0


def test_shard_ranges_cover_everything():
    for mtr, w in [(10000, 8), (100000, 8), (17, 4), (3, 8), (1, 2)]:
        seen = []
        for r in range(w):
            f, c = tspws.shard_range(mtr, r, w)
            seen += list(range(f, f + c))
        assert seen == list(range(mtr))


def test_shard_range_partitions_the_traces(lib):
    """tspws_shard_range (the sharded tspws_main, csrc/comm.hip) and the Python orchestration's shard_range cut an ensemble the
    same way: contiguous, in order, covering every trace once -- also when there are fewer traces than devices."""
    for mtr, n in [(10000, 8), (100000, 8), (7, 3), (3, 8), (0, 4), (12345, 5), (1, 1)]:
        nxt = 0
        for r in range(n):
            f, c = C.c_size_t(), C.c_size_t()
            lib.tspws_shard_range(mtr, r, n, C.byref(f), C.byref(c))
            assert (f.value, c.value) == tspws.shard_range(mtr, r, n)
            assert f.value == nxt
            nxt += c.value
        assert nxt == mtr


def test_comm_needs_a_device(lib):
    """No HIP device: the communicator and the sharded call fail loudly like everything else (no CPU stand-in)."""
    if lib.tspws_hip_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    assert lib.tspws_hip_comm_create(C.byref(h), 2, None) == 5 and not h.value
    assert lib.tspws_hip_multi_create(C.byref(h), 2, None, -1, 3, 4, 2048, 2.0, 1.0, 5.336, 0) == 5 and not h.value


def test_device_list_request_that_needs_one_device_names_the_first_listed(lib, monkeypatch, capfd):
    """A several-device request (TSPWS_DEVICES) whose call needs the whole ensemble on ONE device -- random subsampling, convergence
    curves -- runs on the FIRST device of the list, not on TSPWS_DEVICE's default 0 (csrc/host/tspws_main.c: run_call).  Without such
    a device the call fails with 5 and says which device it wanted."""
    if lib.tspws_hip_device_count() > 5:
        pytest.skip("device 5 exists here")
    monkeypatch.setenv("TSPWS_DEVICES", "5,6")
    monkeypatch.delenv("TSPWS_DEVICE", raising=False)
    X = abi.synth_traces(8, 256, seed=2)
    r = abi.run_main(lib.tspws_main, abi.default_params(subsmpl_N=2, subsmpl_p=0.5), X)
    assert r["rc"] == 5
    C.CDLL(None).fflush(None)                            # (the library prints through libc's stdout buffer)
    out = capfd.readouterr().out
    assert "no usable HIP device 5" in out, out
