"""The front-end's own SAC / msacs I/O (ts-pws_amd/csrc/host/sacio_min.c) against the reference's shipped
example files (three of them are committed under tests/golden/sac as input fixtures).  CPU only."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

import abi

tspws = importlib.import_module("ts-pws_amd")
SACDIR = os.path.join(abi.ROOT, "tests", "golden", "sac")


class SacHeader(C.Structure):
    _fields_ = [("f", C.c_float * 70), ("i", C.c_int32 * 40), ("k", C.c_char * 192)]


@pytest.fixture(scope="module")
def sio():
    path = os.path.join(abi.ROOT, "ts-pws_amd", "lib", "libsacio_min.so")
    if not os.path.exists(path):
        tspws.build()
    lib = C.CDLL(path)
    lib.sac_read.argtypes = [C.c_char_p, C.POINTER(SacHeader), C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.sac_write.argtypes = [C.c_char_p, C.POINTER(SacHeader), C.c_void_p]
    lib.sac_reference_time.restype = C.c_long
    lib.sac_reference_time.argtypes = [C.POINTER(SacHeader)]
    return lib


def test_reads_shipped_example_files(sio, golden):
    files = sorted(os.listdir(SACDIR))
    assert len(files) == 3
    for j, name in enumerate(files):
        h = SacHeader()
        n = C.c_int()
        x = np.zeros(16501, np.float32)
        assert sio.sac_read(os.path.join(SACDIR, name).encode(), C.byref(h), x.ctypes.data, 16501, C.byref(n)) == 0
        assert n.value == 16501 and h.i[6] == 6 and h.i[9] == 16501       # SURVEY.md 8c [measured]
        assert h.f[0] == 4.0 and h.f[5] == -33000.0
        np.testing.assert_array_equal(x, golden["example32"]["traces"][j])  # first 32 files of the same list
        # 2010-01-0(j+1) 00:00:00 UTC
        assert sio.sac_reference_time(C.byref(h)) == 1262304000 + 86400 * j


def test_roundtrip_and_byte_order(sio, tmp_path):
    x = abi.synth_traces(1, 1000, seed=3)[0]
    p_le, p_be = str(tmp_path / "a.sac"), str(tmp_path / "b.sac")
    abi.write_sac(p_le, x, 0.5, -10.0, year=2012, jday=366)
    abi.write_sac(p_be, x, 0.5, -10.0, year=2012, jday=366, big_endian=True)
    for p in (p_le, p_be):
        h = SacHeader()
        n = C.c_int()
        y = np.zeros(1000, np.float32)
        assert sio.sac_read(p.encode(), C.byref(h), y.ctypes.data, 1000, C.byref(n)) == 0
        np.testing.assert_array_equal(x, y)
        assert h.f[0] == 0.5 and h.f[5] == -10.0 and n.value == 1000
        assert sio.sac_reference_time(C.byref(h)) == 1356912000  # 2012-12-31 (leap year day 366)
    # write through the library, read back with the independent python reader
    h = SacHeader()
    n = C.c_int()
    y = np.zeros(1000, np.float32)
    sio.sac_read(p_le.encode(), C.byref(h), y.ctypes.data, 1000, C.byref(n))
    out = str(tmp_path / "c.sac")
    assert sio.sac_write(out.encode(), C.byref(h), y.ctypes.data) == 0
    r = abi.read_sac(out)
    np.testing.assert_array_equal(r["data"], x)
    assert r["f"][1] == x.min() and r["f"][2] == x.max() and r["f"][6] == np.float32(-10.0 + 999 * 0.5)
    # not a SAC file / missing file
    bad = tmp_path / "bad.sac"
    bad.write_bytes(b"\x07" * 700)
    assert sio.sac_read(str(bad).encode(), C.byref(h), None, 0, C.byref(n)) == -3
    assert sio.sac_read(b"/nonexistent.sac", C.byref(h), None, 0, C.byref(n)) == -1


def test_cli_fails_loudly_without_gpu(tmp_path):
    if tspws.load().tspws_hip_device_count() > 0:
        pytest.skip("a HIP device is present")
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(os.path.join(SACDIR, f) for f in sorted(os.listdir(SACDIR))) + "\n")
    exe = os.path.join(abi.ROOT, "ts-pws_amd", "bin", "ts_pws")
    r = subprocess.run([exe, str(lst)], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 5 and "no usable HIP device" in r.stdout
    assert not (tmp_path / "ts_pws.sac").exists()
