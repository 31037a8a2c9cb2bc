"""On-disk formats either side of the path, host side of the C-ABI: Janus FMAT/FVEC and GSL raw blocks
(btk/matrix/gslmatrix.cc:27-96,133-240), HTK parameter files (btk/feature/feature.cc:4025-4318).  The expected bytes are built here, field by
field, from the reference's readers/writers (struct.pack) -- independent of the library's own writers."""
import ctypes as C
import os
import struct

import numpy as np
import pytest


def _load(dsr, path, old, rows, cols):
    L = dsr.load(); m = np.zeros((rows, cols), np.float32); r = C.c_int(); c = C.c_int()
    dsr.check(L.dsr_fmat_load(path.encode(), int(old), rows, cols, m.ctypes.data_as(C.c_void_p), C.byref(r), C.byref(c)))
    return m, r.value, c.value


def test_fmat_reader_all_branches(dsr, tmp_path):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((5, 7)).astype(np.float32)
    be = lambda a: a.astype(">f4").tobytes()
    p = str(tmp_path / "a.fmat")
    open(p, "wb").write(b"FMAT" + struct.pack(">ii", 5, 7) + struct.pack(">f", 0.0) + be(A))
    m, r, c = _load(dsr, p, True, 5, 7)
    assert (r, c) == (5, 7) and np.array_equal(m, A)
    # "number of rows wasn't set" (gslmatrix.cc:63): size1 < 0 -> rows from the file length
    open(p, "wb").write(b"FMAT" + struct.pack(">ii", -1, 7) + struct.pack(">f", 3.0) + be(A))
    m, r, c = _load(dsr, p, True, 5, 7)
    assert (r, c) == (5, 7) and np.array_equal(m, A)
    # a smaller matrix into a larger one: the reference shrinks its matrix to the file's size; data row by row with the file's column count
    B = A[:3, :4].copy()
    open(p, "wb").write(b"FMAT" + struct.pack(">ii", 3, 4) + struct.pack(">f", 0.0) + be(B))
    m, r, c = _load(dsr, p, True, 5, 7)
    assert (r, c) == (3, 4) and np.array_equal(m.reshape(-1)[:12].reshape(3, 4), B)
    # error branches, with the reference's error classes: JIO (8) / JDIMENSION (5)
    for body, status in ((b"FMAX" + struct.pack(">ii", 5, 7) + struct.pack(">f", 0.0) + be(A), 8),        # "Couldn't find magic number in file"
                         (b"FMAT" + struct.pack(">ii", 5, 7) + struct.pack(">f", 0.0), 8),                # "File empty, matrix unchanged!"
                         (b"FMAT" + struct.pack(">ii", 5, 7) + struct.pack(">f", 0.0) + be(A)[:-4], 8),   # "Number of bytes in file = don't match matrix dimension"
                         (b"FMAT" + struct.pack(">ii", 6, 7) + struct.pack(">f", 0.0) + be(np.zeros((6, 7))), 5),   # "Cannot resize from 5 to 6"
                         (b"FMAT" + struct.pack(">ii", 5, 8) + struct.pack(">f", 0.0) + be(np.zeros((5, 8))), 5)):
        open(p, "wb").write(body)
        with pytest.raises(dsr.DsrError) as e:
            _load(dsr, p, True, 5, 7)
        assert e.value.status == status
    with pytest.raises(dsr.DsrError) as e:
        _load(dsr, str(tmp_path / "missing"), True, 5, 7)
    assert e.value.status == 8
    # the GSL raw block (old == False): native floats, exactly rows * cols of them
    open(p, "wb").write(A.tobytes())
    m, r, c = _load(dsr, p, False, 5, 7)
    assert np.array_equal(m, A)
    open(p, "wb").write(A.tobytes()[:-8])
    with pytest.raises(dsr.DsrError) as e:
        _load(dsr, p, False, 5, 7)
    assert e.value.status == 8
    # the library's own writers produce exactly these bytes
    L = dsr.load()
    dsr.check(L.dsr_fmat_save(p.encode(), 1, 5, 7, A.ctypes.data_as(C.c_void_p), 0))
    assert open(p, "rb").read() == b"FMAT" + struct.pack(">ii", 5, 7) + struct.pack(">f", 0.0) + be(A)
    dsr.check(L.dsr_fmat_save(p.encode(), 1, 5, 7, A.ctypes.data_as(C.c_void_p), 1))
    assert open(p, "rb").read()[4:8] == struct.pack(">i", -1)


def test_fvec(dsr, tmp_path):
    L = dsr.load(); v = np.arange(9, dtype=np.float32) * 0.5; p = str(tmp_path / "v")
    open(p, "wb").write(b"FVEC" + struct.pack(">i", 9) + struct.pack(">f", 1.0) + v.astype(">f4").tobytes())
    out = np.zeros(12, np.float32); n = C.c_int()
    dsr.check(L.dsr_fvec_load(p.encode(), 1, 12, out.ctypes.data_as(C.c_void_p), C.byref(n)))
    assert n.value == 9 and np.array_equal(out[:9], v)
    with pytest.raises(dsr.DsrError) as e:
        dsr.check(L.dsr_fvec_load(p.encode(), 1, 4, out.ctypes.data_as(C.c_void_p), C.byref(n)))      # "Cannot resize from 4 to 9"
    assert e.value.status == 5
    open(p, "wb").write(b"FVEC" + struct.pack(">i", 9) + struct.pack(">f", 1.0) + v.astype(">f4").tobytes()[:-4])
    with pytest.raises(dsr.DsrError) as e:
        dsr.check(L.dsr_fvec_load(p.encode(), 1, 12, out.ctypes.data_as(C.c_void_p), C.byref(n)))
    assert e.value.status == 1                                                                          # j_error "Number of bytes ... don't match vector dimension"
    dsr.check(L.dsr_fvec_save(p.encode(), 1, 9, v.ctypes.data_as(C.c_void_p)))
    assert open(p, "rb").read() == b"FVEC" + struct.pack(">i", 9) + struct.pack(">f", 0.0) + v.astype(">f4").tobytes()


def test_htk_parameter_files(dsr, tmp_path):
    """header {int32 nSamples, int32 sampPeriod, int16 sampSize, int16 parmKind} + vectors; isBigEndian == 0 swaps every field
    (feature.cc:4055-4168), which gives the big-endian file HTK itself writes"""
    L = dsr.load(); X = np.random.default_rng(1).standard_normal((6, 13)).astype(np.float32); p = str(tmp_path / "x.htk")
    dsr.check(L.dsr_htk_write(p.encode(), 6, 100000, 52, 6, 0, X.ctypes.data_as(C.c_void_p)))                     # 6 = MFCC
    assert open(p, "rb").read() == struct.pack(">iihh", 6, 100000, 52, 6) + X.astype(">f4").tobytes()
    ns, sp, ss, pk = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    dsr.check(L.dsr_htk_read_header(p.encode(), 0, C.byref(ns), C.byref(sp), C.byref(ss), C.byref(pk)))
    assert (ns.value, sp.value, ss.value, pk.value) == (6, 100000, 52, 6)
    Y = np.zeros_like(X); dsr.check(L.dsr_htk_read(p.encode(), 0, Y.ctypes.data_as(C.c_void_p), X.size))
    assert np.array_equal(X, Y)
    dsr.check(L.dsr_htk_write(p.encode(), 6, 100000, 52, 6, 1, X.ctypes.data_as(C.c_void_p)))                     # "machine is big endian": no swap
    assert open(p, "rb").read() == struct.pack("<iihh", 6, 100000, 52, 6) + X.tobytes()
    for kind in (0o2000 | 6, 0o10000 | 6):                                                                          # _C compressed, _K CRC: refused
        with pytest.raises(dsr.DsrError) as e:
            dsr.check(L.dsr_htk_write(p.encode(), 6, 100000, 52, kind, 0, X.ctypes.data_as(C.c_void_p)))
        assert e.value.status == 8
        with pytest.raises(dsr.DsrError):
            dsr.check(L.dsr_htk_read_header(p.encode(), 0, C.byref(ns), C.byref(sp), C.byref(ss), C.byref(pk)))
