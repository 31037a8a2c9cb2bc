"""ctypes binding of libdsr_hip.so (include/dsr.h).

This is plumbing: device memory and streams come from torch (ROCm), every compute call goes
through the C-ABI with raw device pointers.  There is no CPU path here; if the shared library or
a HIP device is missing the call raises.
"""
import ctypes as C
import os
import re

C_ = C

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(os.path.dirname(_HERE), "lib", "libdsr_hip.so")
_lib = None

vp, i32, i64, f32, f64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint32

ERROR_NAMES = ["OK", "JERROR", "JALLOCATION", "JARITHMETIC", "JCONSISTENCY", "JDIMENSION", "JINDEX", "JINITIALIZATION",
               "JIO", "JITERATOR", "JPYTHON", "JKEY", "JNUMERIC", "JPARAMETER", "JPARSE", "JTYPE"]
E_ITERATOR, E_IO, E_INDEX, E_DIMENSION = 9, 8, 6, 5
E_CONSISTENCY, E_KEY, E_PARAMETER, E_PARSE = 4, 11, 13, 14


class DsrError(Exception):
    """j_error (btk/common/jexception.h:57-70): carries the reference's error_type as .code."""

    def __init__(self, status, msg):
        super().__init__("%s: %s" % (ERROR_NAMES[status] if 0 <= status < len(ERROR_NAMES) else status, msg))
        self.status = status
        self.code = status - 1


class MfccCfg(C.Structure):
    _fields_ = [("blockLen", C.c_int), ("shiftLen", C.c_int), ("padZeros", C.c_int), ("mu", f64), ("fftLen", C.c_int),
                ("powN", C.c_int), ("vtlnRatio", f64), ("vtlnEdge", f64), ("vtlnVersion", C.c_int), ("rate", f32), ("low", f32),
                ("up", f32), ("filterN", C.c_int), ("melVersion", C.c_int), ("logM", f64), ("logA", f64), ("sphinxFlooring", C.c_int),
                ("ncep", C.c_int), ("dctType", C.c_int), ("cmnMode", C.c_int), ("devNormFactor", f64), ("delta", C.c_int),
                ("outDim", C.c_int)]


class DecoderCfg(C.Structure):
    _fields_ = [("beam", f64), ("lmScale", f64), ("lmPenalty", f64), ("silPenalty", f64), ("silenceX", u32),
                ("maxActive", C.c_int), ("maxCandidates", C.c_int), ("arenaTokens", i64), ("streams", C.c_int), ("topN", C.c_int), ("latticeTokens", i64),
                ("wordTrace", C.c_int), ("propagateN", C.c_int), ("fastHash", C.c_int), ("insertSilence", C.c_int), ("wordTraceLattice", C.c_int), ("wordTraces", i64)]


class DecodeResult(C.Structure):
    _fields_ = [("score", f64), ("ac", f32), ("lm", f32), ("frames", i32), ("reachedFinal", i32), ("nArcs", i32),
                ("nWords", i32), ("status", i32), ("maxActiveSeen", i32), ("activeHypos", i64), ("placements", i64), ("registerFrames", i64), ("finalStatesN", i32), ("reserved_", i32)]


_HEADER = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "dsr.h")
_SCALARS = {"int": C.c_int, "dsr_status": C.c_int, "unsigned": C.c_uint, "unsigned int": C.c_uint, "uint32_t": C.c_uint32, "int32_t": C.c_int32,
            "int64_t": C.c_int64, "size_t": C.c_size_t, "float": C.c_float, "double": C.c_double, "uint8_t": C.c_uint8}


def _ctype_of(decl):
    """ctypes type of one C parameter / return declaration of include/dsr.h (pointers are opaque: c_void_p; const char* is c_char_p)."""
    d = re.sub(r"/\*.*?\*/", " ", decl).strip()
    if "*" in d or "[" in d:
        return C.c_char_p if re.match(r"^const\s+char\s*\*\s*\w*$", d) else C.c_void_p
    d = re.sub(r"\bconst\b", " ", d).strip()
    toks = d.split()
    for k in (2, 1):                                   # "unsigned int x", "int x", or a bare type
        for cand in (" ".join(toks[:k]),):
            if cand in _SCALARS and len(toks) <= k + 1:
                return _SCALARS[cand]
    raise ValueError("include/dsr.h: cannot map parameter %r" % decl)


def header_prototypes(path=None):
    """{name: (restype, [argtypes])} for every function include/dsr.h declares -- the single source of the ctypes signatures, so that no
    entry point is ever called with libffi's default int promotion (a 64-bit stride passed as a 32-bit int reads stack garbage)."""
    text = open(path or _HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"\benum\s*\{.*?\}\s*;", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(dsr_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or not ret:
            continue
        restype = None if ret == "void" else _ctype_of(ret)
        argtypes = [] if args in ("", "void") else [_ctype_of(a) for a in args.split(",")]
        out[name] = (restype, argtypes)
    return out


def declare_from_header(L):
    for name, (restype, argtypes) in header_prototypes().items():
        fn = getattr(L, name)                          # AttributeError here = a declared symbol the library does not export
        fn.restype = restype; fn.argtypes = argtypes


def load():
    """Load the library.  torch is imported first so that its HIP runtime (same SONAME) is the one bound."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (binds libamdhip64.so.7 before our library asks for it)
    path = _LIBPATH
    if os.environ.get("DSR_LIB_VARIANT"):      # A/B tooling (tools/ab_*.sh): a variant build under lib/var/<name>/, never copied over the shipped library
        path = os.path.join(os.path.dirname(_LIBPATH), "var", os.environ["DSR_LIB_VARIANT"], "libdsr_hip.so")
    if not os.path.exists(path):
        raise ImportError("libdsr_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` (%s)" % path)
    L = C.CDLL(path)
    declare_from_header(L)
    _lib = L
    return L


def check(status):
    if status != 0:
        raise DsrError(status, (_lib.dsr_last_error() or b"").decode(errors="replace"))


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(a):
    return a.ctypes.data_as(vp)


def _dev(t):
    return C.c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------------------------
class FilterBank:
    """OverSampledDFTAnalysisBank / OverSampledDFTSynthesisBank plan (btk/modulated/modulated.h:291-360)."""

    def __init__(self, prototype, M, m, r, synthesis=False, delayCompensationType=0, gainFactor=1):
        L = load()
        p = _np(prototype, np.float64)
        if p.size != M * m:
            raise DsrError(4, "Prototype sizes do not match (%d vs. %d)." % (p.size, M * m))   # modulated.cc:268-270
        self.h = vp()
        check(L.dsr_fb_create(_ptr(p), M, m, r, int(synthesis), delayCompensationType, gainFactor, C.byref(self.h)))
        self.M, self.m, self.r, self.D, self.synthesis = M, m, r, M >> r, synthesis
        self.pd = L.dsr_fb_processing_delay(self.h)

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_fb_destroy(self.h)

    def frames(self, nsamp):
        return _lib.dsr_fb_analysis_frames(self.h, int(nsamp))

    def blocks(self, nframes):
        return _lib.dsr_fb_synthesis_blocks(self.h, int(nframes))

    def analysis(self, x, nsamp=None):
        """x: cuda float32 [U][C][N] -> complex64 [U][C][Tmax][M/2+1]"""
        import torch
        U, Cn, N = x.shape
        if nsamp is None:
            nsamp = torch.full((U,), N, dtype=torch.int32, device=x.device)
        Tmax = max(1, max(self.frames(int(n)) for n in nsamp.tolist()))
        X = torch.empty((U, Cn, Tmax, self.M // 2 + 1, 2), dtype=torch.float32, device=x.device)
        check(_lib.dsr_fb_analysis(self.h, _dev(x), _dev(nsamp), U, Cn, N, Tmax, _dev(X), cur_stream()))
        return torch.view_as_complex(X)

    def analysis_beamform(self, bf, x, nsamp=None):
        """analysis bank + fixed-weight beamformer in one pass (dsr_fb_analysis_beamform): x cuda float32 [U][C][N] -> complex64 [U][Tmax][M/2+1]"""
        import torch
        U, Cn, N = x.shape
        if nsamp is None:
            nsamp = torch.full((U,), N, dtype=torch.int32, device=x.device)
        Tmax = max(1, max(self.frames(int(n)) for n in nsamp.tolist()))
        Y = torch.empty((U, Tmax, self.M // 2 + 1, 2), dtype=torch.float32, device=x.device)
        check(_lib.dsr_fb_analysis_beamform(self.h, bf.h, _dev(x), _dev(nsamp), U, Cn, N, Tmax, _dev(Y), cur_stream()))
        return torch.view_as_complex(Y)

    def analysis_beamform_supported(self, bf):
        return bool(_lib.dsr_fb_analysis_beamform_supported(self.h, bf.h))

    def synthesis_run(self, Y, nframes=None):
        """Y: cuda complex64 [U][Tmax][M/2+1] -> float32 [U][nblocks*D]"""
        import torch
        U, Tmax, F = Y.shape
        if nframes is None:
            nframes = torch.full((U,), Tmax, dtype=torch.int32, device=Y.device)
        nb = max(1, max(self.blocks(int(n)) for n in nframes.tolist()))
        y = torch.zeros((U, nb * self.D), dtype=torch.float32, device=Y.device)
        Yr = torch.view_as_real(Y.contiguous())
        check(_lib.dsr_fb_synthesis(self.h, _dev(Yr), _dev(nframes), U, Tmax, nb * self.D, _dev(y), cur_stream()))
        return y


class FilterBankState:
    """what a filter bank carries from one block of a long stream to the next (dsr_fb_state): m*M - D samples per (stream, channel) for an
    analysis plan, R*m - 1 subband frames per stream for a synthesis plan."""

    def __init__(self, fb, U, C=1):
        L = load(); self.h = vp(); self.fb, self.U, self.C = fb, U, C
        check(L.dsr_fb_state_create(fb.h, U, C, C_.byref(self.h)))

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_fb_state_destroy(self.h)

    def reset(self):
        check(_lib.dsr_fb_state_reset(self.h))

    def analysis_block(self, x, nsamp=None, last=False):
        """x: cuda float32 [U][C][N] = the block's new samples -> complex64 [U][C][T][M/2+1], T = the frames this block yields"""
        import torch
        U, Cn, N = x.shape
        ns = [N] * U if nsamp is None else [int(v) for v in nsamp]
        nd = torch.tensor(ns, dtype=torch.int32, device=x.device)
        T = max(1, max(_lib.dsr_fb_analysis_block_frames(self.fb.h, self.h, n, int(last)) for n in ns))
        X = torch.empty((U, Cn, T, self.fb.M // 2 + 1, 2), dtype=torch.float32, device=x.device)
        check(_lib.dsr_fb_analysis_block(self.fb.h, self.h, _dev(x), _dev(nd), U, Cn, N, int(last), T, _dev(X), cur_stream()))
        return torch.view_as_complex(X)

    def synthesis_block(self, Y, nframes=None):
        """Y: cuda complex64 [U][T][M/2+1] = the block's new subband frames -> float32 [U][nblocks * D]"""
        import torch
        U, T, F = Y.shape
        nf = [T] * U if nframes is None else [int(v) for v in nframes]
        nd = torch.tensor(nf, dtype=torch.int32, device=Y.device)
        nb = max(1, max(_lib.dsr_fb_synthesis_block_blocks(self.fb.h, self.h, n) for n in nf))
        y = torch.zeros((U, nb * self.fb.D), dtype=torch.float32, device=Y.device)
        Yr = torch.view_as_real(Y.contiguous())
        check(_lib.dsr_fb_synthesis_block(self.fb.h, self.h, _dev(Yr), _dev(nd), max(nf), U, T, nb * self.fb.D, _dev(y), cur_stream()))
        return y


class Beamformer:
    """beamformerWeights + SubbandDS/GSC/MVDR weight design and apply (btk/beamformer/beamformer.h)."""

    def __init__(self, fftLen, chanN, halfBandShift=False):
        L = load()
        self.h = vp(); self.M, self.C = fftLen, chanN
        check(L.dsr_bf_create(fftLen, chanN, int(halfBandShift), C.byref(self.h)))

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_bf_destroy(self.h)

    def calcArrayManifoldVectors(self, sampleRate, delays):
        d = _np(delays, np.float64)
        if d.size != self.C:
            raise DsrError(5, "Number of delays does not match number of channels (%d vs. %d)." % (d.size, self.C))
        check(_lib.dsr_bf_calc_array_manifold(self.h, sampleRate, _ptr(d)))

    def setDiffuseNoiseModel(self, micPositions, sampleRate, sspeed=343740.0):
        mp = _np(micPositions, np.float64)
        check(_lib.dsr_bf_set_diffuse_noise_model(self.h, _ptr(mp), sampleRate, sspeed))

    def divideAllNonDiagonalElements(self, myu):
        check(_lib.dsr_bf_divide_nondiagonal(self.h, myu))

    def setAllLevelsOfDiagonalLoading(self, w):
        check(_lib.dsr_bf_diagonal_loading(self.h, w))

    def setNoiseSpatialSpectralMatrix(self, fbinX, Rnn):
        R = _np(Rnn, np.complex128)
        check(_lib.dsr_bf_set_noise_matrix(self.h, fbinX, _ptr(R)))

    def calcMVDRWeights(self, sampleRate, dThreshold=1.0e-8):
        check(_lib.dsr_bf_calc_mvdr_weights(self.h, sampleRate, dThreshold))

    def calcGSCWeights(self, sampleRate, delays):
        d = _np(delays, np.float64)
        check(_lib.dsr_bf_calc_gsc_weights(self.h, sampleRate, _ptr(d)))

    def setActiveWeights_f(self, fbinX, packedWeight):
        w = _np(packedWeight, np.float64)
        if w.size != 2 * (self.C - 1):
            raise DsrError(5, "the size of an active weight vector must be %d but it is %d" % (2 * (self.C - 1), w.size))
        check(_lib.dsr_bf_set_active_weights(self.h, fbinX, _ptr(w)))

    def zeroActiveWeights(self):
        check(_lib.dsr_bf_zero_active_weights(self.h))

    def select(self, mode):
        check(_lib.dsr_bf_select(self.h, {"ds": 0, "mvdr": 1, "gsc": 2, "gsc_norm": 3, "mvdr_gsc": 4}.get(mode, mode)))

    def get(self, kind):
        F = self.M // 2 + 1
        shape = {0: (self.M, self.C), 1: (F, self.C), 2: (F, self.C, self.C), 3: (self.M, self.C, self.C - 1), 4: (_lib.dsr_bf_bins(self.h), self.C)}[kind]
        out = np.zeros(shape, np.complex128)
        check(_lib.dsr_bf_get(self.h, kind, _ptr(out), out.size * 2))
        return out

    # SubbandMVDRGSC (beamformer.h:394-425): the MVDR vector as quiescent weight, active weights from outside (select("mvdr_gsc"))
    def calcBlockingMatrix1(self, sampleRate, delaysT):
        self.calcGSCWeights(sampleRate, delaysT); return True

    def calcBlockingMatrix2(self):
        try:
            check(_lib.dsr_bf_calc_blocking_matrix2(self.h)); return True
        except DsrError:
            return False                                       # "You have to call calcMVDRWeights() first": the reference returns false

    def upgradeBlockingMatrix(self):
        check(_lib.dsr_bf_upgrade_blocking_matrix(self.h))

    def blockingMatrixOutput(self, X, outChanX=0):
        """X: cuda complex64 [U][C][T][F] -> B[:, outChanX]^H X, [U][T][F]"""
        import torch
        U, Cn, T, F = X.shape
        Y = torch.empty((U, T, F, 2), dtype=torch.float32, device=X.device)
        check(_lib.dsr_bf_blocking_matrix_output(self.h, _dev(torch.view_as_real(X.contiguous())), U, T, int(outChanX), _dev(Y), cur_stream()))
        return torch.view_as_complex(Y)

    # SubbandGSCRLS (beamformer.h:213-262): recursive-least-squares adaptation of the active weights
    def rlsConfig(self, myu=0.9, sigma2=0.0):
        check(_lib.dsr_bf_rls_config(self.h, myu, sigma2))

    def initPrecisionMatrix(self, sigma2=0.01):
        check(_lib.dsr_bf_rls_init_precision(self.h, sigma2))

    def setPrecisionMatrix(self, fbinX, Pz):
        p = np.ascontiguousarray(Pz, np.complex128)
        if p.shape != (self.C - 1, self.C - 1):
            raise DsrError(5, "the precision matrix must be %d x %d" % (self.C - 1, self.C - 1))
        check(_lib.dsr_bf_rls_set_precision(self.h, fbinX, _ptr(p)))

    def setQuadraticConstraint(self, alpha, qctype=1):
        check(_lib.dsr_bf_rls_quadratic_constraint(self.h, alpha, qctype))

    def updateActiveWeightVecotrs(self, flag):
        check(_lib.dsr_bf_rls_adapt(self.h, int(bool(flag))))

    def gsc_rls(self, X, nframes=None):
        """X: cuda complex64 [U][C][T][F] -> (Y [U][T][F], final active weights [U][F][C-1] complex128); nframes: cuda int32 [U] valid frames"""
        import torch
        U, Cn, T, F = X.shape
        Y = torch.empty((U, T, F, 2), dtype=torch.float32, device=X.device)
        wa = torch.zeros((U, F, Cn - 1), dtype=torch.complex128, device=X.device)
        check(_lib.dsr_bf_gsc_rls(self.h, _dev(torch.view_as_real(X.contiguous())), _dev(nframes) if nframes is not None else None, U, T, _dev(Y), _dev(wa), cur_stream()))
        return torch.view_as_complex(Y), wa

    def rlsCarry(self, on=True):
        """keep adapting from call to call (block streaming; the reference's behaviour across reset())"""
        check(_lib.dsr_bf_rls_carry(self.h, int(bool(on))))

    def rlsResetState(self):
        check(_lib.dsr_bf_rls_reset_state(self.h))

    def bins(self):
        """bins per frame of apply(): fftLen/2+1, or fftLen with halfBandShift"""
        return _lib.dsr_bf_bins(self.h)

    def apply(self, X):
        """X: cuda complex64 [U][C][T][F] -> [U][T][F]"""
        import torch
        U, Cn, T, F = X.shape
        Y = torch.empty((U, T, F, 2), dtype=torch.float32, device=X.device)
        check(_lib.dsr_bf_apply(self.h, _dev(torch.view_as_real(X.contiguous())), U, T, _dev(Y), cur_stream()))
        return torch.view_as_complex(Y)


def calcDelaysPolar2(azimuth, elevation, micPositions):
    load()
    mp = _np(micPositions, np.float64); d = np.zeros(mp.shape[0], np.float64)
    check(_lib.dsr_calc_delays_polar2(azimuth, elevation, _ptr(mp), mp.shape[0], _ptr(d)))
    return d


class PrFilterBank:
    """PerfectReconstructionFFTAnalysisBank / SynthesisBank (modulated.cc:686-970), prototype of length 2M*m."""

    def __init__(self, prototype, M, m, r=0):
        L = load(); self.h = vp(); self.M, self.m, self.r = M, m, r
        p = _np(prototype, np.float64)
        if p.size != 2 * M * m:
            raise DsrError(4, "Prototype sizes do not match (%d vs. %d)." % (p.size, 2 * M * m))
        check(L.dsr_prfb_create(_ptr(p), M, m, r, C.byref(self.h)))

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_prfb_destroy(self.h)

    def analysis(self, x, nsamp=None):
        """x: cuda float32 [U][C][N] -> complex64 [U][C][T][2M]"""
        import torch
        U, Cn, N = x.shape
        if nsamp is None:
            nsamp = torch.full((U,), N, dtype=torch.int32, device=x.device)
        T = max(1, max(_lib.dsr_prfb_analysis_frames(self.h, int(n)) for n in nsamp.tolist()))
        X = torch.zeros((U, Cn, T, 2 * self.M), dtype=torch.complex64, device=x.device)
        check(_lib.dsr_prfb_analysis(self.h, _dev(x), _dev(nsamp), U, Cn, N, T, _dev(X), cur_stream()))
        return X

    def synthesis(self, Y, nframes=None):
        """Y: cuda complex64 [U][T][2M] -> float32 [U][(T-(2m-1))*D]"""
        import torch
        U, T, M2 = Y.shape
        if nframes is None:
            nframes = torch.full((U,), T, dtype=torch.int32, device=Y.device)
        nb = max(1, max(_lib.dsr_prfb_synthesis_blocks(self.h, int(n)) for n in nframes.tolist()))
        D = self.M >> self.r
        y = torch.zeros((U, nb * D), dtype=torch.float32, device=Y.device)
        check(_lib.dsr_prfb_synthesis(self.h, _dev(Y.contiguous()), _dev(nframes), U, T, nb * D, _dev(y), cur_stream()))
        return y


class NormalFFTBank:
    """NormalFFTAnalysisBank (modulated.cc:121-257): windowed STFT, windowType 0 rectangle / 1 Hamming / 2 Hanning."""

    def __init__(self, M, r, windowType=1):
        L = load(); self.h = vp(); self.M = M
        check(L.dsr_stft_create(M, r, windowType, C.byref(self.h)))

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_stft_destroy(self.h)

    def frames(self, nsamp):
        return _lib.dsr_stft_frames(self.h, int(nsamp))

    def analysis(self, x, nsamp=None):
        """x: cuda float32 [U][C][N] -> complex64 [U][C][T][M]"""
        import torch
        U, Cn, N = x.shape
        if nsamp is None:
            nsamp = torch.full((U,), N, dtype=torch.int32, device=x.device)
        T = max(1, max(self.frames(int(n)) for n in nsamp.tolist()))
        X = torch.zeros((U, Cn, T, self.M), dtype=torch.complex64, device=x.device)
        check(_lib.dsr_stft_analysis(self.h, _dev(x), _dev(nsamp), U, Cn, N, T, _dev(X), cur_stream()))
        return X


def wpe_single(Y, fftLen, lowerN, upperN, iterationsN=2, loadDb=-20.0, bandWidth=0.0, sampleRate=16000.0, nframes=None, want_filters=False, gn=None):
    """Single-channel WPE (dereverberation.cc:28-300): Y cuda complex64 [U][N][M/2+1] -> out (and the filters [U][M/2+1][P] complex128).
    gn: the filters of the utterance / block before (reset() keeps them in the reference): used as the start and overwritten."""
    import torch
    load()
    U, N, F = Y.shape
    if nframes is None:
        nframes = torch.full((U,), N, dtype=torch.int32, device=Y.device)
    out = torch.zeros((U, N, F), dtype=torch.complex64, device=Y.device)
    if gn is not None:
        check(_lib.dsr_wpe_single_continue(_dev(Y.contiguous()), _dev(nframes), U, N, fftLen, lowerN, upperN, iterationsN, loadDb, bandWidth, sampleRate, _dev(out),
                                           _dev(gn), cur_stream()))
        return out, gn
    gn = torch.zeros((U, F, upperN - lowerN + 1), dtype=torch.complex128, device=Y.device) if want_filters else None
    check(_lib.dsr_wpe_single(_dev(Y.contiguous()), _dev(nframes), U, N, fftLen, lowerN, upperN, iterationsN, loadDb, bandWidth, sampleRate, _dev(out),
                              _dev(gn) if want_filters else None, cur_stream()))
    return (out, gn) if want_filters else out


def wpe_multi(Y, fftLen, lowerN, upperN, iterationsN=2, loadDb=-20.0, bandWidth=0.0, sampleRate=16000.0, nframes=None, filterChan=-1):
    """Multi-channel WPE (dereverberation.cc:281-586): Y cuda complex64 [U][C][N][M/2+1] -> (out, filters [U][C][M/2+1][C*P] complex128).
    filterChan >= 0: all channels through that channel's filter (the reference's getOutput when that channel's feature pulls first)."""
    import torch
    load()
    U, Cn, N, F = Y.shape
    if nframes is None:
        nframes = torch.full((U,), N, dtype=torch.int32, device=Y.device)
    out = torch.zeros((U, Cn, N, F), dtype=torch.complex64, device=Y.device)
    gn = torch.zeros((U, Cn, F, Cn * (upperN - lowerN + 1)), dtype=torch.complex128, device=Y.device)
    check(_lib.dsr_wpe_multi(_dev(Y.contiguous()), _dev(nframes), U, Cn, N, fftLen, lowerN, upperN, iterationsN, loadDb, bandWidth, sampleRate, int(filterChan),
                             _dev(out), _dev(gn), cur_stream()))
    return out, gn


class ZelinskiPostFilter:
    """Zelinski post-filter (postfilter.cc:8-221,350-493); manifold [M/2+1][C] complex = arrayManifold() (or wq() with type | 8)."""

    def __init__(self, fftLen, chanN, manifold, alpha=0.6, type=2, minFrames=0):
        L = load(); self.h = vp(); self.M, self.C = fftLen, chanN
        check(L.dsr_zelinski_create(fftLen, chanN, alpha, type, minFrames, C.byref(self.h)))
        m = np.ascontiguousarray(manifold, np.complex128)
        for f in range(fftLen // 2 + 1):
            check(L.dsr_zelinski_set_manifold(self.h, f, _ptr(m[f])))

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_zelinski_destroy(self.h)

    def apply(self, X, Y, nframes=None, want_weights=False):
        """X: cuda complex64 [U][C][T][F], Y: [U][T][F] -> out [U][T][F] (and the weights [U][T][F] fp32)"""
        import torch
        U, Cn, T, F = X.shape
        if nframes is None:
            nframes = torch.full((U,), T, dtype=torch.int32, device=X.device)
        out = torch.zeros((U, T, F), dtype=torch.complex64, device=X.device)
        w = torch.zeros((U, T, F), dtype=torch.float32, device=X.device) if want_weights else None
        check(_lib.dsr_zelinski_apply(self.h, _dev(X.contiguous()), _dev(Y.contiguous()), _dev(nframes), U, T, _dev(out), _dev(w) if want_weights else None, cur_stream()))
        return (out, w) if want_weights else out

    def apply_bf(self, bf, X, nframes=None, want_weights=False, want_bf_output=False):
        """The post-filter behind its beamformer (setBeamformer, postfilter.cc:376): out = postfilter(X, bf(X)); the beamformer's sum is formed in the filter's own
        pass over the snapshots where it streams them.  X: cuda complex64 [U][C][T][F] -> out [U][T][F] (+ the weights, + bf(X))"""
        import torch
        U, Cn, T, F = X.shape
        if nframes is None:
            nframes = torch.full((U,), T, dtype=torch.int32, device=X.device)
        out = torch.zeros((U, T, F), dtype=torch.complex64, device=X.device)
        w = torch.zeros((U, T, F), dtype=torch.float32, device=X.device) if want_weights else None
        Y = torch.zeros((U, T, F), dtype=torch.complex64, device=X.device) if want_bf_output else None
        check(_lib.dsr_zelinski_apply_bf(self.h, bf.h, _dev(X.contiguous()), _dev(nframes), U, T, _dev(out), _dev(w) if want_weights else None,
                                         _dev(Y) if want_bf_output else None, cur_stream()))
        r = (out,) + ((w,) if want_weights else ()) + ((Y,) if want_bf_output else ())
        return r if len(r) > 1 else out

    def carry(self, on=True):
        """keep the spectral densities from call to call (block streaming)"""
        check(_lib.dsr_zelinski_carry(self.h, int(bool(on))))

    def resetState(self):
        check(_lib.dsr_zelinski_reset_state(self.h))


class SubbandMMI:
    """SubbandMMI (beamformer.h:264-312, beamformer.cc:1753-2319) over dsr_mmi_*; the reference's method names."""

    def __init__(self, fftLen=512, chanN=2, halfBandShift=False, targetSourceX=0, nSource=2, pfType=0, alpha=0.9):
        L = load(); self.h = vp(); self.M, self.C, self.nSource, self.NC = fftLen, chanN, nSource, 1
        check(L.dsr_mmi_create(fftLen, chanN, int(bool(halfBandShift)), targetSourceX, nSource, pfType, alpha, C.byref(self.h)))

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_mmi_destroy(self.h)

    def bins(self):
        return _lib.dsr_mmi_bins(self.h)

    def outBins(self):
        """bins per output frame: bins(), or fftLen with the APAB post-filter (whose frames are not conjugate-symmetric)"""
        return _lib.dsr_mmi_out_bins(self.h)

    def useBinaryMask(self, avgFactor=-1.0, fwidth=1, type=0):
        check(_lib.dsr_mmi_use_binary_mask(self.h, avgFactor, fwidth, type))

    def calcWeights(self, sampleRate, delays):
        d = np.ascontiguousarray(delays, np.float64)
        if d.shape != (self.nSource, self.C):
            raise DsrError(5, "delays must be [%d][%d]" % (self.nSource, self.C))
        check(_lib.dsr_mmi_calc_weights(self.h, sampleRate, _ptr(d))); self.NC = 1

    def calcWeightsN(self, sampleRate, delays, NC=2):
        d = np.ascontiguousarray(delays, np.float64)
        if d.shape != (self.nSource, self.C):
            raise DsrError(5, "delays must be [%d][%d]" % (self.nSource, self.C))
        check(_lib.dsr_mmi_calc_weights_n(self.h, sampleRate, _ptr(d), NC)); self.NC = NC

    def setActiveWeights_f(self, fbinX, packedWeights, option=0):
        w = np.ascontiguousarray(packedWeights, np.float64)
        if w.ndim != 2:
            raise DsrError(5, "packedWeights must be a matrix")
        check(_lib.dsr_mmi_set_active_weights_f(self.h, fbinX, _ptr(w), w.shape[0], w.shape[1], option))

    def setHiActiveWeights_f(self, fbinX, pkdWa, pkdwb, option=0):
        a = np.ascontiguousarray(pkdWa, np.float64).ravel(); b = np.ascontiguousarray(pkdwb, np.float64).ravel()
        check(_lib.dsr_mmi_set_hi_active_weights_f(self.h, fbinX, _ptr(a), a.size, _ptr(b), b.size, option))

    def get(self, srcX, kind):
        """kind: 'wq' [M][C], 'wl' [M][C], 'B' [M][C][C-NC], 'ta' [M][C], 'wa' [M][C-NC] (complex128)"""
        k = {"wq": 0, "wl": 1, "B": 2, "ta": 3, "wa": 4}[kind]; bs = self.C - self.NC
        shape = {0: (self.M, self.C), 1: (self.M, self.C), 2: (self.M, self.C, bs), 3: (self.M, self.C), 4: (self.M, bs)}[k]
        out = np.zeros(shape, np.complex128)
        check(_lib.dsr_mmi_get(self.h, srcX, k, _ptr(out), 2 * out.size))
        return out

    def apply(self, X, nframes=None):
        """X: cuda complex64 [U][C][T][bins] -> [U][T][outBins]"""
        import torch
        U, Cn, T, F = X.shape
        if Cn != self.C or F != self.bins():
            raise DsrError(5, "snapshots must be [U][%d][T][%d]" % (self.C, self.bins()))
        if nframes is None:
            nframes = torch.full((U,), T, dtype=torch.int32, device=X.device)
        out = torch.zeros((U, T, self.outBins()), dtype=torch.complex64, device=X.device)
        check(_lib.dsr_mmi_apply(self.h, _dev(X.contiguous()), _dev(nframes), U, T, _dev(out), cur_stream()))
        return out


class McCowanPostFilter(ZelinskiPostFilter):
    """McCowan post-filter (postfilter.cc:502-945): Zelinski's recursions + a noise coherence matrix per bin."""

    def __init__(self, fftLen, chanN, manifold, alpha=0.6, type=2, minFrames=0, threshold=0.99):
        L = load(); self.h = vp(); self.M, self.C = fftLen, chanN
        check(L.dsr_mccowan_create(fftLen, chanN, alpha, type, minFrames, threshold, C.byref(self.h)))
        m = np.ascontiguousarray(manifold, np.complex128)
        for f in range(fftLen // 2 + 1):
            check(L.dsr_zelinski_set_manifold(self.h, f, _ptr(m[f])))

    def setDiffuseNoiseModel(self, micPositions, sampleRate, sspeed=343740.0):
        mp = _np(micPositions, np.float64); check(_lib.dsr_mccowan_set_diffuse_noise_model(self.h, _ptr(mp), sampleRate, sspeed)); return True

    def setNoiseSpatialSpectralMatrix(self, fbinX, Rnn):
        r = np.ascontiguousarray(Rnn, np.complex128); check(_lib.dsr_mccowan_set_noise_matrix(self.h, fbinX, _ptr(r))); return True

    def setAllLevelsOfDiagonalLoading(self, w):
        check(_lib.dsr_mccowan_diagonal_loading(self.h, -1, w))

    def setLevelOfDiagonalLoading(self, fbinX, w):
        check(_lib.dsr_mccowan_diagonal_loading(self.h, fbinX, w))

    def divideAllNonDiagonalElements(self, myu):
        check(_lib.dsr_mccowan_divide_nondiagonal(self.h, myu))


class LefkimmiatisPostFilter(McCowanPostFilter):
    """Lefkimmiatis post-filter (postfilter.cc:948-1210): McCowan's clean-signal estimate against the coherence-based noise estimate,
    divided by d^H pinv(R) d from bin fbinX1 on."""

    def __init__(self, fftLen, chanN, manifold, minSV=1e-8, fbinX1=0, alpha=0.6, type=2, minFrames=0, threshold=0.99):
        L = load(); self.h = vp(); self.M, self.C = fftLen, chanN
        check(L.dsr_lefkimmiatis_create(fftLen, chanN, minSV, fbinX1, alpha, type, minFrames, threshold, C.byref(self.h)))
        m = np.ascontiguousarray(manifold, np.complex128)
        for f in range(fftLen // 2 + 1):
            check(L.dsr_zelinski_set_manifold(self.h, f, _ptr(m[f])))


class LpcEnvelope:
    """WarpMVDR/BurgMVDR (kind 0) and WarpLPC/BurgLPC (kind 1) envelopes of windowed frames, lpc.h:134-195,291-331."""

    def __init__(self, dim, order=60, warp=0.0, method=0, kind=0, correlate=0):
        L = load(); self.h = vp(); self.dim = dim
        check(L.dsr_lpc_create(dim, order, correlate, warp, method, kind, C.byref(self.h)))

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_lpc_destroy(self.h)

    def run(self, frames):
        """frames: cuda float32 [T][dim] -> float64 [T][dim/2+1]"""
        import torch
        T, dim = frames.shape
        assert dim == self.dim
        out = torch.zeros((T, dim // 2 + 1), dtype=torch.float64, device=frames.device)
        check(_lib.dsr_lpc_run(self.h, _dev(frames), T, _dev(out), cur_stream()))
        return out


class Mfcc:
    def __init__(self, lda=None, **kw):
        L = load()
        self.cfg = MfccCfg(); L.dsr_mfcc_default_cfg(C.byref(self.cfg))
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        if lda is None and "outDim" not in kw:
            self.cfg.outDim = 0
        a = _np(lda, np.float32) if lda is not None else None
        self.h = vp()
        check(L.dsr_mfcc_create(C.byref(self.cfg), _ptr(a) if a is not None else None, C.byref(self.h)))
        self.outDim = L.dsr_mfcc_out_dim(self.h)

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_mfcc_destroy(self.h)

    def frames(self, nsamp):
        return _lib.dsr_mfcc_frames(self.h, int(nsamp))

    def run(self, y, nsamp=None, stage=0):
        """y: cuda float32 [U][N] -> float32 [U][Tmax][dim]"""
        import torch
        U, N = y.shape
        if nsamp is None:
            nsamp = torch.full((U,), N, dtype=torch.int32, device=y.device)
        c = self.cfg
        raw = lambda n: ((n + c.shiftLen - 1) // c.shiftLen) if c.padZeros else max(0, -(-(n - c.blockLen) // c.shiftLen))
        Tmax = max(1, max(raw(int(n)) for n in nsamp.tolist()))
        dim = {0: self.outDim, 1: c.ncep, 2: c.ncep, 3: c.filterN, 4: c.powN}[stage]
        out = torch.zeros((U, Tmax, dim), dtype=torch.float32, device=y.device)
        check(_lib.dsr_mfcc_run(self.h, _dev(y), _dev(nsamp), U, N, Tmax, stage, _dev(out), cur_stream()))
        return out


class Gmm:
    def __init__(self, refN=None, mean=None, ivar=None, det=None, val=None, scale=None, files=None):
        L = load(); self.h = vp()
        if files is not None:
            check(L.dsr_gmm_load(files[0].encode(), files[1].encode(), C.byref(self.h)))
        else:
            r = _np(refN, np.int32); m = _np(mean, np.float32); iv = _np(ivar, np.float32); d = _np(det, np.float32); v = _np(val, np.float32)
            s = _np(scale, np.float32) if scale is not None else None
            check(L.dsr_gmm_create(len(r), m.shape[1], _ptr(r), _ptr(m), _ptr(iv), _ptr(d), _ptr(v), _ptr(s) if s is not None else None, C.byref(self.h)))
        self.K = L.dsr_gmm_num_dists(self.h); self.D = L.dsr_gmm_dim(self.h)

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_gmm_destroy(self.h)

    def save(self, cbFile, dsFile):
        check(_lib.dsr_gmm_save(self.h, cbFile.encode(), dsFile.encode()))

    def score(self, x, mode=0, want_argmin=True):
        """x: cuda float32 [N][D] -> (score [N][K] float32, argmin [N][K] uint8)"""
        import torch
        N = x.shape[0]
        sc = torch.empty((N, self.K), dtype=torch.float32, device=x.device)
        am = torch.zeros((N, self.K), dtype=torch.uint8, device=x.device) if want_argmin and mode != 1 else None
        check(_lib.dsr_gmm_score(self.h, _dev(x), N, mode, _dev(sc), _dev(am) if am is not None else None, cur_stream()))
        return sc, am


class Wfst:
    """WFSTFlyWeight (asr/decoder/wfstFlyWeight.h:47-119)."""

    def __init__(self):
        L = load(); self.h = vp(); check(L.dsr_wfst_create(C.byref(self.h)))

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_wfst_destroy(self.h)

    def read(self, fileName, binary=False):
        check(_lib.dsr_wfst_read(self.h, fileName.encode(), int(binary)))

    def read_dynamic(self, fileName, noSelfLoops=False):
        """WFSTransducer::read (asr/fsm/fsm.cc:901-986)"""
        check(_lib.dsr_wfst_read_dynamic(self.h, fileName.encode(), int(noSelfLoops)))

    def write(self, fileName, binary=True):
        check(_lib.dsr_wfst_write(self.h, fileName.encode(), int(binary)))

    def add_arc(self, s1, s2, i, o, cost=0.0):
        check(_lib.dsr_wfst_add_arc(self.h, s1, s2, i, o, cost))

    def add_final(self, s, cost=0.0):
        check(_lib.dsr_wfst_add_final(self.h, s, cost))

    def export(self):
        n = _lib.dsr_wfst_num_nodes(self.h); a = _lib.dsr_wfst_num_arcs(self.h)
        d = dict(nodeState=np.zeros(n, np.uint32), nodeFinal=np.zeros(n, np.int32), nodeCost=np.zeros(n, np.float32),
                 arcOff=np.zeros(n + 1, np.int32), arcDst=np.zeros(max(a, 1), np.int32), arcIn=np.zeros(max(a, 1), np.uint32),
                 arcOut=np.zeros(max(a, 1), np.uint32), arcCost=np.zeros(max(a, 1), np.float32))
        check(_lib.dsr_wfst_export(self.h, *[_ptr(d[k]) for k in ("nodeState", "nodeFinal", "nodeCost", "arcOff", "arcDst", "arcIn", "arcOut", "arcCost")]))
        for k in ("arcDst", "arcIn", "arcOut", "arcCost"):
            d[k] = d[k][:a]
        return d


class Decoder:
    """DecoderFlyWeight (asr/decoder/decoder.i:147-199) for batches of score matrices."""

    def __init__(self, beam=100.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silenceX=0xFFFFFFFF, maxActive=0,
                 maxCandidates=0, arenaTokens=0, streams=0, latticeTokens=0, topN=0, wordTrace=0, generateLattice=True, propagateN=5, fastHash=False,
                 insertSilence=False, wordTraces=0):
        L = load(); c = DecoderCfg(); L.dsr_decoder_default_cfg(C.byref(c))
        c.wordTrace, c.wordTraceLattice, c.propagateN, c.fastHash, c.insertSilence, c.wordTraces = int(wordTrace), int(generateLattice), propagateN, int(fastHash), int(insertSilence), wordTraces
        c.beam, c.lmScale, c.lmPenalty, c.silPenalty, c.silenceX = beam, lmScale, lmPenalty, silPenalty, silenceX
        c.maxActive, c.maxCandidates, c.arenaTokens, c.streams, c.latticeTokens, c.topN = maxActive, maxCandidates, arenaTokens, streams, latticeTokens, topN
        self.h = vp(); check(L.dsr_decoder_create(C.byref(c), C.byref(self.h))); self._g = None

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_decoder_destroy(self.h)

    def set(self, wfst):
        check(_lib.dsr_decoder_set(self.h, wfst.h)); self._g = wfst

    def setBeam(self, beam):
        check(_lib.dsr_decoder_set_beam(self.h, beam))

    def enable_dump(self, on=True):
        check(_lib.dsr_decoder_enable_dump(self.h, int(on)))

    def decode_batch(self, scores, nframes=None, maxPath=None):
        """scores: cuda float32 [U][T][nDist].  Returns list of dicts."""
        import torch
        U, T, nDist = scores.shape
        if nframes is None:
            nframes = torch.full((U,), T, dtype=torch.int32, device=scores.device)
        if maxPath is None:
            maxPath = 4 * T + 64
        res = (DecodeResult * U)()
        arcs = np.zeros((U, maxPath), np.int32); words = np.zeros((U, maxPath), np.uint32)
        check(_lib.dsr_decoder_decode_batch(self.h, _dev(scores), _dev(nframes), U, T, nDist, C.byref(res), _ptr(arcs), _ptr(words), maxPath, cur_stream()))
        out = []
        for u in range(U):
            r = res[u]
            out.append(dict(status=r.status, score=r.score, ac=r.ac, lm=r.lm, frames=r.frames, reachedFinal=bool(r.reachedFinal),
                            arcs=arcs[u, :min(r.nArcs, maxPath)].copy(), words=words[u, :min(r.nWords, maxPath)].copy(),
                            activeHypos=r.activeHypos, maxActive=r.maxActiveSeen, placements=r.placements, registerFrames=r.registerFrames, finalStatesN=r.finalStatesN))
        return out

    def lattice(self, u=0, eosX=0):
        """_Decoder::lattice() (decoder.h:805-860) of utterance u of the last decode (needs latticeTokens > 0)"""
        h = vp(); check(_lib.dsr_decoder_lattice(self.h, int(u), int(eosX), C.byref(h)))
        return Lattice(h)

    def writeGMM(self, u, conv, channel, spk, utt, cfrom, score, fileName="", frameInterval=0.01):
        """_Decoder::writeGMM (decoder.h:1018-1102) of utterance u of the last decode (needs latticeTokens > 0 and set_symbols)"""
        check(_lib.dsr_decoder_write_gmm(self.h, int(u), conv.encode(), channel.encode(), spk.encode(), utt.encode(), float(cfrom), float(score),
                                         (fileName or "").encode(), float(frameInterval)))

    def get_dump(self):
        n = i64(); fo = C.POINTER(i64)(); nd = C.POINTER(i32)(); ac = C.POINTER(f32)(); lm = C.POINTER(f32)(); arc = C.POINTER(i32)()
        check(_lib.dsr_decoder_get_dump(self.h, C.byref(n), C.byref(fo), C.byref(nd), C.byref(ac), C.byref(lm), C.byref(arc)))
        nf = n.value
        off = np.array([fo[i] for i in range(nf + 1)], np.int64) if nf > 0 else np.zeros(1, np.int64)
        N = int(off[-1])
        g = lambda p, dt: np.ctypeslib.as_array(p, (N,)).astype(dt).copy() if N > 0 else np.zeros(0, dt)
        return dict(frameOff=off, node=g(nd, np.int32), ac=g(ac, np.float32), lm=g(lm, np.float32), arc=g(arc, np.int32))


def _lexh(lex):
    """handle of a lexicon object (asr.dictionary.LexiconPtr keeps it in _h) or None"""
    if lex is None:
        return None
    return getattr(lex, "_h", None) or getattr(lex, "h", None)


class Lattice:
    """asr/lattice Lattice as the decoder builds it: nodes numbered as the reference numbers them (0 = initial), edges in creation order."""

    def __init__(self, handle):
        self.h = handle
        n, e = _lib.dsr_lattice_num_nodes(self.h), _lib.dsr_lattice_num_edges(self.h)
        d = {"nodeFinal": np.zeros(n, np.int32), "from": np.zeros(e, np.int32), "to": np.zeros(e, np.int32), "in": np.zeros(e, np.uint32),
             "out": np.zeros(e, np.uint32), "start": np.zeros(e, np.int32), "end": np.zeros(e, np.int32), "ac": np.zeros(e, np.float64), "lm": np.zeros(e, np.float64)}
        check(_lib.dsr_lattice_get(self.h, *[_ptr(d[k]) for k in ("nodeFinal", "from", "to", "in", "out", "start", "end", "ac", "lm")]))
        self.data = d; self.finalStatesN = _lib.dsr_lattice_final_states_n(self.h)

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_lattice_destroy(self.h)

    def write(self, fileName, useSymbols=False, writeData=False):
        """Lattice::write (lattice.cc:715-757); useSymbols needs the lexica and is not offered here"""
        if useSymbols:
            raise DsrError(13, "useSymbols: write the numeric form and map the symbols with the lexica")
        check(_lib.dsr_lattice_write(self.h, fileName.encode(), int(writeData)))

    # ---- asr/lattice operations (lattice.i:79-123); symbols cross the boundary as indices, asr/lattice.py joins them with the lexica
    @staticmethod
    def read(fileName, noSelfLoops=False, readData=False, inlex=None, outlex=None):
        """WFST::read(fileName, noSelfLoops, readData) (fsm.h:3787-3873); inlex/outlex: Lexicon objects (or None) for symbolic files"""
        load(); h = vp()
        check(_lib.dsr_lattice_read(fileName.encode(), int(noSelfLoops), int(readData), _lexh(inlex), _lexh(outlex), C.byref(h)))
        return Lattice(h)

    def rescore(self, lmScale=30.0, lmPenalty=0.0, silPenalty=0.0, silenceX=0):
        sc = C.c_float(0.0)
        check(_lib.dsr_lattice_rescore(self.h, float(lmScale), float(lmPenalty), float(silPenalty), int(silenceX), C.byref(sc)))
        return np.float32(sc.value)

    def bestHypo(self, useInputSymbols=False):
        n = C.c_int(0); check(_lib.dsr_lattice_best_hypo(self.h, int(useInputSymbols), None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint32)
        check(_lib.dsr_lattice_best_hypo(self.h, int(useInputSymbols), _ptr(out), out.size, C.byref(n)))
        return out[:n.value].copy()

    def gammaProbs(self, acScale=1.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silenceX=0):
        p = C.c_double(0.0)
        check(_lib.dsr_lattice_gamma_probs(self.h, float(acScale), float(lmScale), float(lmPenalty), float(silPenalty), int(silenceX), C.byref(p)))
        return p.value

    def gammaProbsDist(self, distribset, acScale=1.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silenceX=0):
        """Lattice::gammaProbsDist (lattice.cc:331-341): distribset = a dsr_distribset handle (asr.gaussian.DistribSetBasicPtr keeps it in _ds)"""
        p = C.c_double(0.0)
        check(_lib.dsr_lattice_gamma_probs_dist(self.h, distribset, float(acScale), float(lmScale), float(lmPenalty), float(silPenalty), int(silenceX), C.byref(p)))
        return p.value

    def prune(self, threshold=100.0):
        check(_lib.dsr_lattice_prune(self.h, float(threshold)))

    def pruneEdges(self, edgesN=0):
        check(_lib.dsr_lattice_prune_edges(self.h, int(edgesN)))

    def purge(self):
        check(_lib.dsr_lattice_purge(self.h))

    def state(self):
        """per link (creation order): gamma, still on its node's list; per node (creation order): printed index, still held, forward, backward"""
        n, e = _lib.dsr_lattice_num_nodes(self.h), _lib.dsr_lattice_num_edges(self.h)
        d = dict(gamma=np.zeros(e, np.float64), edgeLive=np.zeros(e, np.int32), nodeIndex=np.zeros(n, np.int32), nodeLive=np.zeros(n, np.int32),
                 fwd=np.zeros(n, np.float64), bwd=np.zeros(n, np.float64))
        check(_lib.dsr_lattice_get_state(self.h, *[_ptr(d[k]) for k in ("gamma", "edgeLive", "nodeIndex", "nodeLive", "fwd", "bwd")]))
        return d

    def writeCTM(self, outlex, conv, channel, spk, utt, cfrom, score, fileName="", frameInterval=0.01, endMarker="</s>"):
        check(_lib.dsr_lattice_write_ctm(self.h, _lexh(outlex), conv.encode(), channel.encode(), spk.encode(), utt.encode(), float(cfrom), float(score),
                                         fileName.encode(), float(frameInterval), endMarker.encode()))

    def writePhoneCTM(self, inlex, conv, channel, spk, utt, cfrom, score, fileName="", frameInterval=0.01, endMarker="</s>"):
        check(_lib.dsr_lattice_write_phone_ctm(self.h, _lexh(inlex), conv.encode(), channel.encode(), spk.encode(), utt.encode(), float(cfrom), float(score),
                                               fileName.encode(), float(frameInterval), endMarker.encode()))

    def writeHypoHTK(self, outlex, conv, channel, spk, utt, cfrom, score, fileName="", flag=0, frameInterval=0.01, endMarker="</s>"):
        check(_lib.dsr_lattice_write_hypo_htk(self.h, _lexh(outlex), conv.encode(), channel.encode(), spk.encode(), utt.encode(), float(cfrom), float(score),
                                              fileName.encode(), int(flag), float(frameInterval), endMarker.encode()))

    def writeWordConfs(self, outlex, fileName, uttId, endMarker="</s>"):
        check(_lib.dsr_lattice_write_word_confs(self.h, _lexh(outlex), fileName.encode(), uttId.encode(), endMarker.encode()))

    def pack(self):
        n = _lib.dsr_lattice_pack_size(self.h); b = np.zeros(n, np.uint8)
        check(_lib.dsr_lattice_pack(self.h, _ptr(b), n)); return b

    @staticmethod
    def unpack(buf):
        load(); b = np.ascontiguousarray(buf, np.uint8); h = vp()
        check(_lib.dsr_lattice_unpack(_ptr(b), b.size, C.byref(h))); return Lattice(h)


class Pipe:
    def __init__(self, ana, syn, bf, mfcc, gmm, dec, gmmMode=0, fused=False):
        """fused: analysis bank and fixed-weight beamformer as one kernel where supported (the channel snapshots are then never written)"""
        L = load(); self.h = vp(); self._keep = (ana, syn, bf, mfcc, gmm, dec)
        check(L.dsr_pipe_create(ana.h, syn.h, bf.h, mfcc.h, gmm.h, dec.h, gmmMode, C.byref(self.h)))
        if fused:
            check(L.dsr_pipe_set_fused(self.h, 1))

    def __del__(self):
        if _lib is not None and getattr(self, "h", None):
            _lib.dsr_pipe_destroy(self.h)

    def run(self, x, nsamp_dev, nsamp_host, maxPath=4096, want_paths=True):
        U, Cn, N = x.shape
        res = (DecodeResult * U)()
        ns = _np(nsamp_host, np.int32)
        arcs = np.zeros((U, maxPath), np.int32) if want_paths else None
        words = np.zeros((U, maxPath), np.uint32) if want_paths else None
        check(_lib.dsr_pipe_run(self.h, _dev(x), _dev(nsamp_dev), _ptr(ns), U, Cn, N, C.byref(res),
                                _ptr(arcs) if want_paths else None, _ptr(words) if want_paths else None, maxPath, cur_stream()))
        return res, arcs, words

    def submit(self, x, nsamp_dev, nsamp_host, maxPath=4096, want_paths=True):
        """Enqueue one batch on the current stream and return; collect() waits for it.  One batch in flight per Pipe."""
        U, Cn, N = x.shape
        ns = _np(nsamp_host, np.int32)
        self._inflight = (U, maxPath, want_paths, x, nsamp_dev, ns)              # keep the inputs alive until collected
        check(_lib.dsr_pipe_submit(self.h, _dev(x), _dev(nsamp_dev), _ptr(ns), U, Cn, N, maxPath, 1 if want_paths else 0, cur_stream()))

    def collect(self, reuse=False):
        """reuse: hand out the same host arrays call after call (rows are valid up to nArcs / nWords, the rest is whatever the call before left):
        a fresh zero-filled pair is 16 MB of page faults per 1000-utterance batch"""
        U, maxPath, want_paths = self._inflight[:3]
        res = (DecodeResult * U)()
        if reuse and want_paths:
            h = getattr(self, "_host", None)
            if h is None or h[0].shape != (U, maxPath):
                h = self._host = (np.zeros((U, maxPath), np.int32), np.zeros((U, maxPath), np.uint32))
            arcs, words = h
        else:
            arcs = np.zeros((U, maxPath), np.int32) if want_paths else None
            words = np.zeros((U, maxPath), np.uint32) if want_paths else None
        check(_lib.dsr_pipe_collect(self.h, C.byref(res), _ptr(arcs) if want_paths else None, _ptr(words) if want_paths else None))
        self._inflight = None
        return res, arcs, words

    def stage_ms(self):
        ms = (f32 * 6)(); check(_lib.dsr_pipe_stage_ms(self.h, ms)); return list(ms)

    def intermediate(self, which):
        p = vp(); n = i64(); check(_lib.dsr_pipe_intermediate(self.h, which, C.byref(p), C.byref(n))); return p.value, n.value

    def intermediate_host(self, which, dtype=np.float32):
        p, n = self.intermediate(which)
        out = np.zeros(n // np.dtype(dtype).itemsize, dtype)
        check(_lib.dsr_memcpy_dtoh(_ptr(out), vp(p), n, cur_stream()))
        return out
