"""btk.stream: FeatureStream protocol (btk/stream/stream.h:36-75, stream.i:42-277) over dsr_stream_* handles."""
import ctypes as C

import numpy as np

from .. import _capi as K

_NP = {0: np.int8, 1: np.int16, 2: np.float32, 3: np.float64, 4: np.complex128}


def lib():
    """the loaded library; every signature comes from include/dsr.h (dsr._capi.declare_from_header)"""
    return K.load()


class FeatureStreamPtr(object):
    """Base of every *FeatureStreamPtr: next(frameX=-5) returns a numpy view of the operator's own buffer
    (valid until the next call, btk/include/vector.i:51-65), reset(), size(), name(), current(), isEnd(), frameX()."""

    def __init__(self, handle, keep=()):
        self._h = handle; self._keep = tuple(keep)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().dsr_stream_release(self._h)
        except Exception:
            pass

    def next(self, frameX=-5):
        L = lib(); p = C.c_void_p(); n = C.c_size_t()
        st = L.dsr_stream_next(self._h, int(frameX), C.byref(p), C.byref(n))
        if st == K.E_ITERATOR:
            raise StopIteration                                   # JITERATOR -> StopIteration (jexception.i:178-180)
        if st == K.E_IO:
            raise IOError((L.dsr_last_error() or b"").decode())
        K.check(st)
        dt = _NP[L.dsr_stream_type(self._h)]
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_char)), (n.value * np.dtype(dt).itemsize,)).view(dt)

    __next__ = next

    def __iter__(self):
        self.reset(); return self

    def current(self):
        L = lib(); p = C.c_void_p(); n = C.c_size_t(); K.check(L.dsr_stream_current(self._h, C.byref(p), C.byref(n)))
        dt = _NP[L.dsr_stream_type(self._h)]
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_char)), (n.value * np.dtype(dt).itemsize,)).view(dt)

    def reset(self):
        K.check(lib().dsr_stream_reset(self._h))

    def size(self):
        return lib().dsr_stream_size(self._h)

    def name(self):
        return lib().dsr_stream_name(self._h).decode()

    def isEnd(self):
        return bool(lib().dsr_stream_is_end(self._h))

    def frameX(self):
        return lib().dsr_stream_frameX(self._h)


VectorFloatFeatureStreamPtr = VectorFeatureStreamPtr = VectorComplexFeatureStreamPtr = VectorShortFeatureStreamPtr = FeatureStreamPtr


_REFILL = C.CFUNCTYPE(C.c_int, C.c_void_p)


def _new(fn, *args, keep=()):
    h = C.c_void_p(); K.check(fn(*args, C.byref(h))); return h, keep


class _PySource(FeatureStreamPtr):
    """PyVector*FeatureStreamPtr(iterable): adapts a Python iterable with size()/reset()/__iter__ (pyStream.h:44-152).
    The iterable is drained into a frame source at every reset()."""
    _TYPE = 2
    _exc = None

    def __init__(self, src, name="PyFeatureStream"):
        self._src = src; sz = int(src.size())
        h, _ = _new(lib().dsr_frame_source_create, self._TYPE, sz, name.encode())
        FeatureStreamPtr.__init__(self, h); self._load()
        # a reset() of any downstream operator cascades to this source in C++ (streams.cpp FrameSrc::reset): the refill callback then calls the
        # Python iterable's reset() and drains it again before the next frame is served, as PyFeatureStream::reset does (pyStream.h:100-130)
        self._exc = None
        self._cb = _REFILL(self._refill)
        K.check(lib().dsr_frame_source_set_refill(self._h, C.cast(self._cb, C.c_void_p), None))

    def _refill(self, _user):
        try:
            if hasattr(self._src, "reset"):
                self._src.reset()
            self._load()
            return 0
        except BaseException as e:                                 # JPYTHON: re-raised by next() (jexception.i:181-183)
            self._exc = e
            return 1

    def next(self, frameX=-5):
        try:
            return FeatureStreamPtr.next(self, frameX)
        except K.DsrError as e:
            if e.status == 10 and self._exc is not None:
                exc, self._exc = self._exc, None
                raise exc
            raise

    __next__ = next

    def _load(self):
        dt = _NP[self._TYPE]
        rows = [np.array(v, dtype=dt).reshape(-1) for v in iter(self._src)]
        a = np.ascontiguousarray(np.stack(rows) if rows else np.zeros((0, self.size()), dt))
        K.check(lib().dsr_frame_source_set_frames(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0]))

    def reset(self):
        if hasattr(self._src, "reset"):
            self._src.reset()
        self._load()


class PyVectorShortFeatureStreamPtr(_PySource):
    _TYPE = 1


class PyVectorFloatFeatureStreamPtr(_PySource):
    _TYPE = 2


class PyVectorFeatureStreamPtr(_PySource):
    _TYPE = 3


class PyVectorComplexFeatureStreamPtr(_PySource):
    _TYPE = 4

    def __init__(self, src, name="PyFeatureStream"):
        if isinstance(src, FeatureStreamPtr):                    # testNyquistFilterBankDesign.py:48 wraps a native stream: pass through
            lib().dsr_stream_retain(src._h); FeatureStreamPtr.__init__(self, src._h, keep=(src,))
        else:
            _PySource.__init__(self, src, name)

    def reset(self):
        if self._keep:
            FeatureStreamPtr.reset(self)
        else:
            _PySource.reset(self)
