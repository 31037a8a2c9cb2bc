"""oracle/oracle.py -- TEST INFRASTRUCTURE (never imported by the product).

ctypes/numpy face of oracle/_build/liborc.so, the CPU restatement of the reference's
hot path (see oracle/orc.h for the file:line map).  Importers: tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(native=False):
    tgt = "_build/liborc_native.so" if native else "_build/liborc.so"
    subprocess.check_call(["make", "-s", "-C", _HERE, tgt])
    return os.path.join(_HERE, tgt)


def build_ref():
    """oracle/_ref: the reference's in-tree LINPACK, only where /root/reference exists."""
    if not os.path.isdir("/root/reference/btk/matrix"):
        return None
    subprocess.check_call(["make", "-s", "-C", _HERE, "_ref"])
    return os.path.join(_HERE, "_ref/libref_linpack.so")


def lib(native=False):
    global _LIB
    if native:
        p = os.path.join(_HERE, "_build/liborc_native.so")
        if not os.path.exists(p):
            p = build(native=True)
        return _bind(C.CDLL(p))
    if _LIB is None:
        p = os.path.join(_HERE, "_build/liborc.so")
        if not os.path.exists(p):
            p = build()
        _LIB = _bind(C.CDLL(p))
    return _LIB


def ref_linpack():
    p = os.path.join(_HERE, "_ref/libref_linpack.so")
    if not os.path.exists(p):
        return None
    return C.CDLL(p)


c_int, c_dbl, c_flt, c_vp = C.c_int, C.c_double, C.c_float, C.c_void_p


class MfccCfg(C.Structure):
    _fields_ = [("blockLen", c_int), ("shiftLen", c_int), ("padZeros", c_int), ("mu", c_dbl),
                ("fftLen", c_int), ("powN", c_int), ("vtlnRatio", c_dbl), ("vtlnEdge", c_dbl),
                ("vtlnVersion", c_int), ("rate", c_flt), ("low", c_flt), ("up", c_flt),
                ("filterN", c_int), ("melVersion", c_int), ("logM", c_dbl), ("logA", c_dbl),
                ("ncep", c_int), ("dctType", c_int), ("devNormFactor", c_dbl), ("delta", c_int),
                ("outDim", c_int), ("lda", c_vp), ("sphinxFlooring", c_int)]


class CbSet(C.Structure):
    _fields_ = [("K", c_int), ("dimN", c_int), ("refN", c_vp), ("off", c_vp), ("mean", c_vp),
                ("ivar", c_vp), ("det", c_vp), ("count", c_vp), ("pi", c_vp), ("scale", c_vp)]


class DecCfg(C.Structure):
    _fields_ = [("beam", c_dbl), ("lmScale", c_dbl), ("lmPenalty", c_dbl), ("silPenalty", c_dbl),
                ("silenceX", C.c_uint), ("dumpTokens", c_int), ("topN", c_int)]


class DecResult(C.Structure):
    _fields_ = [("score", c_dbl), ("ac", c_flt), ("lm", c_flt), ("frames", c_int),
                ("reachedFinal", c_int), ("nArcs", c_int), ("arcs", C.POINTER(c_int)),
                ("arcFrames", C.POINTER(c_int)), ("nWords", c_int), ("words", C.POINTER(C.c_uint)),
                ("activeHypos", C.c_long), ("nActive", c_int), ("activeCount", C.POINTER(c_int)),
                ("topScore", C.POINTER(c_dbl)), ("dumpOff", C.POINTER(C.c_long)),
                ("dumpNode", C.POINTER(c_int)), ("dumpAc", C.POINTER(c_flt)),
                ("dumpLm", C.POINTER(c_flt)), ("dumpArc", C.POINTER(c_int)),
                ("dumpN", C.c_long), ("dumpCap", C.c_long), ("finalStatesN", c_int)]


class Lattice(C.Structure):
    _fields_ = [("nNodes", c_int), ("capNodes", c_int), ("nodeFinal", C.POINTER(c_int)), ("nodeFirstEdge", C.POINTER(c_int)),
                ("nEdges", c_int), ("capEdges", c_int), ("from_", C.POINTER(c_int)), ("to", C.POINTER(c_int)), ("in_", C.POINTER(C.c_uint)),
                ("out", C.POINTER(C.c_uint)), ("start", C.POINTER(c_int)), ("end", C.POINTER(c_int)), ("ac", C.POINTER(c_dbl)),
                ("lm", C.POINTER(c_dbl)), ("nextEdge", C.POINTER(c_int))]


def _bind(L):
    L.orc_melbank_create.restype = c_vp
    L.orc_wfst_new.restype = c_vp
    L.orc_cbset_load.restype = C.POINTER(CbSet)
    for n in ("orc_melbank_create",):
        getattr(L, n).argtypes = [c_int, c_flt, c_flt, c_flt, c_int, c_int]
    L.orc_calc_mainlobe.argtypes = [c_dbl, c_vp, c_int, c_int, c_vp]
    L.orc_calc_delays_polar2.argtypes = [c_flt, c_flt, c_vp, c_int, c_vp]
    L.orc_diffuse_noise_model.argtypes = [c_vp, c_int, c_int, c_dbl, c_dbl, c_vp]
    L.orc_divide_nondiag.argtypes = [c_vp, c_int, c_int, c_flt]
    L.orc_diagonal_loading.argtypes = [c_vp, c_int, c_int, c_flt]
    L.orc_pseudoinverse.argtypes = [c_vp, c_int, c_vp, c_flt]
    L.orc_csvdc.argtypes = [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_int]
    L.orc_mvdr_weights.argtypes = [c_vp, c_vp, c_int, c_int, c_dbl, c_vp]
    L.orc_preemphasis.argtypes = [c_vp, c_int, c_int, c_dbl, c_vp]
    L.orc_vtln.argtypes = [c_vp, c_int, c_int, c_dbl, c_dbl, c_int, c_vp]
    L.orc_log.argtypes = [c_vp, c_int, c_int, c_dbl, c_dbl, c_int, c_vp]
    L.orc_cmn_batch.argtypes = [c_vp, c_int, c_int, c_dbl, c_vp, c_vp, c_vp]
    L.orc_cmn_runon.argtypes = [c_vp, c_int, c_int, c_dbl, c_vp]
    L.orc_wfst_add_arc.argtypes = [c_vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, c_flt]
    L.orc_wfst_add_final.argtypes = [c_vp, C.c_uint, c_flt]
    for n in ("orc_wfst_free", "orc_wfst_num_nodes", "orc_wfst_num_arcs"):
        getattr(L, n).argtypes = [c_vp]
    L.orc_wfst_read.argtypes = [c_vp, C.c_char_p, c_int]
    L.orc_wfst_write.argtypes = [c_vp, C.c_char_p, c_int]
    L.orc_wfst_export.argtypes = [c_vp] * 9
    L.orc_decode.argtypes = [c_vp, C.POINTER(DecCfg), c_vp, c_int, c_int, C.POINTER(DecResult)]
    L.orc_melbank_free.argtypes = [c_vp]
    L.orc_mel.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_vp]
    return L


def _p(a):
    return a.ctypes.data_as(c_vp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ------------------------------------------------------------------ filter banks
def analysis_num_frames(nsamp, M, m, r, dctype=0):
    return lib().orc_analysis_num_frames(int(nsamp), M, m, r, dctype)


def analysis_bank(x, h, M, m, r, dctype=0, gain=1):
    """-> complex128 [T][M] (modulated.cc:412-452)."""
    x = _f32(x); h = _f64(h)
    T = analysis_num_frames(len(x), M, m, r, dctype)
    X = np.zeros((T, M, 2), np.float64)
    lib().orc_analysis_bank(_p(x), c_int(len(x)), _p(h), M, m, r, dctype, gain, _p(X))
    return X[..., 0] + 1j * X[..., 1]


def synthesis_bank(Y, g, M, m, r, dctype=0, gain=1):
    """complex [T][M] -> float32 samples (modulated.cc:626-664)."""
    Y = np.ascontiguousarray(Y, dtype=np.complex128); g = _f64(g)
    T = Y.shape[0]; D = M >> r
    out = np.zeros((max(T, 1), D), np.float32)
    n = lib().orc_synthesis_bank(_p(Y.view(np.float64)), T, _p(g), M, m, r, dctype, gain, _p(out))
    return out[:n].reshape(-1)


def pr_analysis_bank(x, h, M, m, r):
    """PerfectReconstructionFFTAnalysisBank (modulated.cc:686-818): [T][2M] complex128."""
    x = _f32(x); h = _f64(h); L = lib()
    T = L.orc_pr_analysis_num_frames(len(x), M, m, r)
    X = np.zeros((T, 2 * M), np.complex128)
    L.orc_pr_analysis_bank(_p(x), len(x), _p(h), M, m, r, _p(X))
    return X


def pr_synthesis_bank(Y, g, M, m, r):
    """PerfectReconstructionFFTSynthesisBank (modulated.cc:820-970): [T-(2m-1)][D] float32."""
    Y = np.ascontiguousarray(Y, np.complex128); g = _f64(g); L = lib()
    T = Y.shape[0]; D = M >> r
    out = np.zeros((max(0, T - (2 * m - 1)), D), np.float32)
    L.orc_pr_synthesis_bank.restype = C.c_int
    n = L.orc_pr_synthesis_bank(_p(Y), T, _p(g), M, m, r, _p(out))
    return out[:n]


def normal_fft_bank(x, M, r, winType=1):
    x = _f32(x)
    T = lib().orc_normal_fft_num_frames(len(x), M, r)
    X = np.zeros((T, M, 2), np.float64)
    lib().orc_normal_fft_bank(_p(x), len(x), M, r, winType, _p(X))
    return X[..., 0] + 1j * X[..., 1]


# ------------------------------------------------------------------ beamformer
def calc_mainlobe(fs, delays, M):
    d = _f64(delays); Cn = len(d)
    wq = np.zeros((M, Cn, 2), np.float64)
    lib().orc_calc_mainlobe(fs, _p(d), Cn, M, _p(wq))
    return wq[..., 0] + 1j * wq[..., 1]


def calc_mainlobe_hbs(fs, delays, M):
    """calcMainlobe with halfBandShift == true (beamformer.cc:544-555)"""
    d = _f64(delays); Cn = len(d)
    wq = np.zeros((M, Cn, 2), np.float64)
    L = lib(); L.orc_calc_mainlobe_hbs.argtypes = [c_dbl, c_vp, c_int, c_int, c_vp]
    L.orc_calc_mainlobe_hbs(fs, _p(d), Cn, M, _p(wq))
    return wq[..., 0] + 1j * wq[..., 1]


def apply_all_bins(X, wq, B=None, wa=None, normalize=False):
    """halfBandShift == true: X [C][T][M], wq [M][C], B [M][C][C-1], wa [M][C-1] -> Y [T][M], every bin on its own (beamformer.cc:1159-1175,1321-1330)"""
    X = np.ascontiguousarray(X, np.complex128); Cn, T, M = X.shape
    wq = np.ascontiguousarray(wq, np.complex128); Y = np.zeros((T, M), np.complex128)
    Bc = np.ascontiguousarray(B, np.complex128) if B is not None else None; wac = np.ascontiguousarray(wa, np.complex128) if wa is not None else None
    L = lib(); L.orc_apply_all_bins.argtypes = [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]
    L.orc_apply_all_bins(_p(X), _p(wq), _p(Bc) if Bc is not None else None, _p(wac) if wac is not None else None, Cn, T, M, int(normalize), _p(Y))
    return Y


def calc_delays_polar2(az, el, micpos):
    mp = _f64(micpos); Cn = mp.shape[0]
    d = np.zeros(Cn, np.float64)
    lib().orc_calc_delays_polar2(az, el, _p(mp), Cn, _p(d))
    return d


def diffuse_noise_model(micpos, M, fs, sspeed=343740.0, mu=None, loading=None):
    mp = _f64(micpos); Cn = mp.shape[0]
    R = np.zeros((M // 2 + 1, Cn, Cn, 2), np.float64)
    lib().orc_diffuse_noise_model(_p(mp), Cn, M, fs, sspeed, _p(R))
    if mu is not None:
        lib().orc_divide_nondiag(_p(R), Cn, M, mu)
    if loading is not None:
        lib().orc_diagonal_loading(_p(R), Cn, M, loading)
    return R[..., 0] + 1j * R[..., 1]


def pseudoinverse(A, thr=1e-8):
    A = np.ascontiguousarray(A, dtype=np.complex128); n = A.shape[0]
    out = np.zeros((n, n), np.complex128)
    ok = lib().orc_pseudoinverse(_p(A.view(np.float64)), n, _p(out.view(np.float64)), thr)
    return out, bool(ok)


def csvdc(A):
    """LINPACK csvdc (job 11) as restated in orc_svd.c: A (n x p) -> (info, s[min(n, p)], U [n][n], V [p][p]) complex64"""
    A = np.asarray(A); n, p = A.shape
    a = np.asfortranarray(A.astype(np.complex64)); s = np.zeros(2 * (n + p) + 2, np.complex64); e = np.zeros(2 * (n + p) + 2, np.complex64)
    u = np.zeros((n, n), np.complex64, order="F"); v = np.zeros((p, p), np.complex64, order="F")
    info = lib().orc_csvdc(_p(a), n, n, p, _p(s), _p(e), _p(u), n, _p(v), p)
    return info, s[:min(n, p)].copy(), u, v


def ref_csvdc(A):
    """the reference's own csvdc through oracle/_ref (authoring container only); same returns as csvdc()"""
    R = ref_linpack()
    A = np.asarray(A); n, p = A.shape
    a = np.asfortranarray(A.astype(np.complex64)); s = np.zeros(2 * (n + p) + 2, np.complex64); e = np.zeros(2 * (n + p) + 2, np.complex64)
    u = np.zeros((n, n), np.complex64, order="F"); v = np.zeros((p, p), np.complex64, order="F")
    info = R.ref_csvdc(_p(a), n, n, p, _p(s), _p(e), _p(u), n, _p(v), p, 11)
    return info, s[:min(n, p)].copy(), u, v


def mvdr_weights(wq, R, thr=1e-8):
    wq = np.ascontiguousarray(wq, dtype=np.complex128); R = np.ascontiguousarray(R, dtype=np.complex128)
    M, Cn = wq.shape
    w = np.zeros((M // 2 + 1, Cn), np.complex128)
    lib().orc_mvdr_weights(_p(wq.view(np.float64)), _p(R.view(np.float64)), Cn, M, thr, _p(w.view(np.float64)))
    return w


def beamform_apply(X, W):
    """X [C][T][M] complex, W [M/2+1][C] -> Y [T][M]."""
    X = np.ascontiguousarray(X, dtype=np.complex128); W = np.ascontiguousarray(W, dtype=np.complex128)
    Cn, T, M = X.shape
    Y = np.zeros((T, M), np.complex128)
    lib().orc_beamform_apply(_p(X.view(np.float64)), _p(W.view(np.float64)), Cn, T, M, _p(Y.view(np.float64)))
    return Y


def blocking_matrix(d):
    d = np.ascontiguousarray(d, dtype=np.complex128); Cn = len(d)
    B = np.zeros((Cn, Cn - 1), np.complex128)
    ok = lib().orc_blocking_matrix(_p(d.view(np.float64)), Cn, _p(B.view(np.float64)))
    return B, bool(ok)


def gsc_apply(X, wq, B, wa, normalize=False):
    X = np.ascontiguousarray(X, dtype=np.complex128)
    Cn, T, M = X.shape
    wq = np.ascontiguousarray(wq, dtype=np.complex128); B = np.ascontiguousarray(B, dtype=np.complex128)
    wa = np.ascontiguousarray(wa, dtype=np.complex128)
    Y = np.zeros((T, M), np.complex128)
    lib().orc_gsc_apply(_p(X.view(np.float64)), _p(wq.view(np.float64)), _p(B.view(np.float64)),
                        _p(wa.view(np.float64)), Cn, T, M, int(normalize), _p(Y.view(np.float64)))
    return Y


def gsc_rls(X, wq, B, myu=0.9, sigma2=0.0, sigma2init=0.01, alpha=-1.0, qctype=0, adapt=True, normalize=False, P0=None):
    """SubbandGSCRLS (beamformer.cc:1497-1698): X [C][T][M], wq [M][C], B [M/2+1][C][C-1] -> (Y [T][M], final wa [M/2+1][C-1]).
    sigma2 = the constructor's diagonal weight, sigma2init = initPrecisionMatrix's argument (P0 overrides: [M/2+1][n][n])."""
    X = np.ascontiguousarray(X, dtype=np.complex128); Cn, T, M = X.shape; n = Cn - 1; F = M // 2 + 1
    wq = np.ascontiguousarray(wq, dtype=np.complex128); B = np.ascontiguousarray(B, dtype=np.complex128)
    if P0 is None:
        P0 = np.broadcast_to(np.eye(n, dtype=np.complex128) * float(np.float32(1.0) / np.float32(sigma2init)), (F, n, n))   # 1/sigma2 is a float division (:1534)
    P0 = np.ascontiguousarray(P0, dtype=np.complex128)
    dw = np.full(F, np.float32(sigma2), np.float64)
    Y = np.zeros((T, M), np.complex128); wa = np.zeros((F, n), np.complex128)
    L = lib(); L.orc_gsc_rls.restype = None
    L.orc_gsc_rls(_p(X), _p(wq), _p(B), _p(P0), _p(dw), Cn, T, M, C.c_double(float(np.float32(myu))), C.c_double(float(np.float32(alpha))), int(qctype), int(bool(adapt)),
                  int(bool(normalize)), _p(Y), _p(wa))
    return Y, wa


class SubbandMMI:
    """SubbandMMI (beamformer.h:264-312, beamformer.cc:1753-2319) over oracle/orc_mmi.c; method names as in the reference."""

    def __init__(self, fftLen=512, halfBandShift=False, targetSourceX=0, nSource=2, pfType=0, alpha=0.9, chanN=None):
        L = lib(); L.orc_mmi_create.restype = C.c_void_p
        self.M, self.Cn, self.nSource, self.hbs, self.NC = fftLen, int(chanN), nSource, bool(halfBandShift), 1
        self.h = C.c_void_p(L.orc_mmi_create(fftLen, int(chanN), int(bool(halfBandShift)), targetSourceX, nSource, pfType, C.c_double(alpha)))

    def __del__(self):
        try:
            L = lib(); L.orc_mmi_free.argtypes = [C.c_void_p]; L.orc_mmi_free(self.h)
        except Exception:
            pass

    def reset(self):
        L = lib(); L.orc_mmi_reset.argtypes = [C.c_void_p]; L.orc_mmi_reset(self.h)

    def useBinaryMask(self, avgFactor=-1.0, fwidth=1, type=0):
        L = lib(); L.orc_mmi_use_binary_mask.argtypes = [C.c_void_p, C.c_double, C.c_uint, C.c_uint]
        L.orc_mmi_use_binary_mask(self.h, avgFactor, fwidth, type)

    def calcWeights(self, sampleRate, delays):
        d = np.ascontiguousarray(delays, np.float64); assert d.shape == (self.nSource, self.Cn)
        L = lib(); L.orc_mmi_calc_weights.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.orc_mmi_calc_weights(self.h, sampleRate, _p(d)); self.NC = 1

    def calcWeightsN(self, sampleRate, delays, NC=2):
        d = np.ascontiguousarray(delays, np.float64); assert d.shape == (self.nSource, self.Cn)
        L = lib(); L.orc_mmi_calc_weights_n.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_uint]
        rc = L.orc_mmi_calc_weights_n(self.h, sampleRate, _p(d), NC)
        if rc != 0:
            raise ValueError("calcWeightsN: rc %d" % rc)
        self.NC = NC

    def setActiveWeights_f(self, fbinX, packedWeights, option=0):
        w = np.ascontiguousarray(packedWeights, np.float64)
        if w.shape != (self.nSource, 2 * (self.Cn - self.NC)):
            raise ValueError("packed weights must be [%d][%d]" % (self.nSource, 2 * (self.Cn - self.NC)))
        L = lib(); L.orc_mmi_set_active_weights_f.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        rc = L.orc_mmi_set_active_weights_f(self.h, fbinX, _p(w), option)
        if rc != 0:
            raise ValueError("setActiveWeights_f: rc %d" % rc)

    def setHiActiveWeights_f(self, fbinX, pkdWa, pkdwb, option=0):
        a = np.ascontiguousarray(pkdWa, np.float64); b = np.ascontiguousarray(pkdwb, np.float64)
        if a.size != 2 * self.nSource * (self.Cn - self.NC) * self.nSource or b.size != 2 * self.nSource * self.nSource:
            raise ValueError("sizes")
        L = lib(); L.orc_mmi_set_hi_active_weights_f.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_int]
        rc = L.orc_mmi_set_hi_active_weights_f(self.h, fbinX, _p(a), _p(b), option)
        if rc != 0:
            raise ValueError("setHiActiveWeights_f: rc %d" % rc)

    def get(self, srcX, kind):
        """kind: 'wq' [M][C], 'wl' [M][C], 'B' [M][C][C-NC], 'ta' [M][C], 'wa' [M][C-NC]"""
        k = {"wq": 0, "wl": 1, "B": 2, "ta": 3, "wa": 4}[kind]; bs = self.Cn - self.NC
        shape = {0: (self.M, self.Cn), 1: (self.M, self.Cn), 2: (self.M, self.Cn, bs), 3: (self.M, self.Cn), 4: (self.M, bs)}[k]
        out = np.zeros(shape, np.complex128)
        L = lib(); L.orc_mmi_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_mmi_get(self.h, srcX, k, _p(out))
        return out

    def run(self, X):
        """X [C][T][Fin] complex (Fin = fftLen with halfBandShift, else >= fftLen/2+1) -> the frames' output vectors [T][fftLen]."""
        X = np.ascontiguousarray(X, np.complex128); Cn, T, Fin = X.shape
        out = np.zeros((T, self.M), np.complex128)
        L = lib(); L.orc_mmi_next.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        for t in range(T):
            xt = np.ascontiguousarray(X[:, t, :])
            rc = L.orc_mmi_next(self.h, _p(xt), Fin, _p(out[t]))
            if rc != 0:
                raise ValueError("SubbandMMI.next: rc %d" % rc)
        return out


# ------------------------------------------------------------------ MFCC chain
def zelinski_postfilter(X, Y, wq, alpha=0.6, type=2, minFrames=0):
    """X [C][T][F], Y [T][F], wq [F][C] complex -> (out [T][F] complex128, wp1 [T][F]) (postfilter.cc:56-221,428-493)."""
    X = np.ascontiguousarray(X, np.complex128); Y = np.ascontiguousarray(Y, np.complex128); wq = np.ascontiguousarray(wq, np.complex128)
    Cn, T, F = X.shape
    out = np.zeros((T, F), np.complex128); wp1 = np.zeros((T, F), np.float64)
    L = lib(); L.orc_zelinski_postfilter.restype = C.c_int
    rc = L.orc_zelinski_postfilter(_p(X), _p(Y), _p(wq), Cn, T, F, C.c_double(alpha), type, minFrames, _p(out), _p(wp1))
    if rc != 0:
        raise ValueError("The number of channels %d is <= 1" % Cn)
    return out, wp1


def pf_diffuse_noise_model(micpos, M, fs, sspeed=343740.0):
    mp = np.ascontiguousarray(micpos, np.float64); Cn = mp.shape[0]
    R = np.zeros((M // 2 + 1, Cn, Cn), np.complex128)
    lib().orc_pf_diffuse_noise_model(_p(mp), Cn, M, C.c_double(fs), C.c_double(sspeed), _p(R))
    return R


def mccowan_postfilter(X, Y, wq, R, alpha=0.6, type=2, minFrames=0, threshold=0.99):
    """McCowanPostFilter (postfilter.cc:568-945): X [C][T][F], Y [T][F], wq [F][C], R [F][C][C] -> (out, wp1)."""
    X = np.ascontiguousarray(X, np.complex128); Y = np.ascontiguousarray(Y, np.complex128); wq = np.ascontiguousarray(wq, np.complex128)
    R = np.ascontiguousarray(R, np.complex128)
    Cn, T, F = X.shape
    out = np.zeros((T, F), np.complex128); wp1 = np.zeros((T, F), np.float64)
    L = lib(); L.orc_mccowan_postfilter.restype = C.c_int
    rc = L.orc_mccowan_postfilter(_p(X), _p(Y), _p(wq), _p(R), Cn, T, F, C.c_double(alpha), type, minFrames, C.c_double(threshold), _p(out), _p(wp1))
    if rc != 0:
        raise ValueError("The number of channels %d is <= 1" % Cn)
    return out, wp1


def lefkimmiatis_lambda(R, d, minSV=1e-8):
    """d^H pinv(R_f) d per bin (LefkimmiatisPostFilter::calcInverseNoiseSpatialSpectralMatrix + calcLambda, postfilter.cc:981-1009) with
    the reference's pseudo-inverse as restated in orc_pseudoinverse (beamformer.cc:253-300: LINPACK csvdc on complex<float>, singular
    values below minSV dropped, pinned against the reference's own csvdc in test_oracle_cpu.py).  A bin whose SVD drops a value keeps
    its pseudo-inverse here; the reference replaces such a matrix by the identity (postfilter.cc:989-991), restated too.
    R [F][C][C], d [F][C] -> [F] complex128."""
    R = np.asarray(R, np.complex128); d = np.asarray(d, np.complex128)
    lam = np.zeros(R.shape[0], np.complex128)
    for f in range(R.shape[0]):
        pinv, ok = pseudoinverse(R[f], minSV)
        if not ok:
            pinv = np.eye(R.shape[1], dtype=np.complex128)         # "if( false == ret ) gsl_matrix_complex_set_identity( _invR[fbinX] )"
        lam[f] = np.vdot(pinv.conj().T @ d[f], d[f])           # tmpH = invR^H d; Lambda = tmpH^H d
    return lam


def lefkimmiatis_postfilter(X, Y, wq, R, lam, alpha=0.6, type=2, minFrames=0, threshold=0.99, fbinX1=0):
    """LefkimmiatisPostFilter (postfilter.cc:948-1210): X [C][T][F], Y [T][F], wq [F][C] (array manifold), R [F][C][C], lam [F] -> (out, wp1)."""
    X = np.ascontiguousarray(X, np.complex128); Y = np.ascontiguousarray(Y, np.complex128); wq = np.ascontiguousarray(wq, np.complex128)
    R = np.ascontiguousarray(R, np.complex128); lam = np.ascontiguousarray(lam, np.complex128)
    Cn, T, F = X.shape
    out = np.zeros((T, F), np.complex128); wp1 = np.zeros((T, F), np.float64)
    L = lib(); L.orc_lefkimmiatis_postfilter.restype = C.c_int
    rc = L.orc_lefkimmiatis_postfilter(_p(X), _p(Y), _p(wq), _p(R), _p(lam), Cn, T, F, C.c_double(alpha), type, minFrames, C.c_double(threshold), int(fbinX1),
                                       _p(out), _p(wp1))
    if rc != 0:
        raise ValueError("The number of channels %d is <= 1" % Cn)
    return out, wp1


def wpe_single(Y, lowerN, upperN, iterationsN=2, loadDb=-20.0, bandWidth=0.0, sampleRate=16000.0, gnInit=None):
    """SingleChannelWPEDereverberationFeature (dereverberation.cc:28-300): Y [N][M] complex -> (out [N][M], gn [M][P]).
    gnInit [M][P]: the filters left by the utterance before (reset() keeps them, nextSpeaker() zeroes them, :258-277)."""
    Y = np.ascontiguousarray(Y, np.complex128); N, M = Y.shape; P = upperN - lowerN + 1
    out = np.zeros((N, M), np.complex128); gn = np.zeros((M, P), np.complex128)
    g0 = np.ascontiguousarray(gnInit, np.complex128) if gnInit is not None else None
    L = lib(); L.orc_wpe_single_w.restype = C.c_int
    rc = L.orc_wpe_single_w(_p(Y), N, M, lowerN, upperN, iterationsN, C.c_double(loadDb), C.c_double(bandWidth), C.c_double(sampleRate),
                            _p(g0) if g0 is not None else None, _p(out), _p(gn))
    if rc != 0:
        raise ValueError("wpe_single failed (%d)" % rc)
    return out, gn


def wpe_multi(Y, lowerN, upperN, iterationsN=2, loadDb=-20.0, bandWidth=0.0, sampleRate=16000.0, filterChan=-1):
    """MultiChannelWPEDereverberation (dereverberation.cc:281-586): Y [C][N][M] complex -> (out [C][N][M], gn [C][M][C*P]).
    filterChan >= 0: all channels through that channel's filter (the reference's getOutput when that channel's feature pulls first)."""
    Y = np.ascontiguousarray(Y, np.complex128); Cn, N, M = Y.shape; P = upperN - lowerN + 1
    out = np.zeros((Cn, N, M), np.complex128); gn = np.zeros((Cn, M, Cn * P), np.complex128)
    L = lib(); L.orc_wpe_multi.restype = C.c_int
    rc = L.orc_wpe_multi(_p(Y), Cn, N, M, lowerN, upperN, iterationsN, C.c_double(loadDb), C.c_double(bandWidth), C.c_double(sampleRate), int(filterChan),
                         _p(out), _p(gn))
    if rc != 0:
        raise ValueError("wpe_multi failed (%d)" % rc)
    return out, gn


def lpc_feature(frames, order, warp=0.0, method=0, kind=0):
    """WarpMVDR/BurgMVDR (kind 0) and WarpLPC/BurgLPC (kind 1) spectral envelopes, lpc.h:134-195,291-331."""
    L = lib(); fr = _f32(frames); T, dim = fr.shape
    out = np.zeros((T, dim // 2 + 1), np.float64)
    L.orc_lpc_feature.restype = C.c_int
    rc = L.orc_lpc_feature(_p(fr), C.c_long(T), dim, order, C.c_float(warp), method, kind, _p(out))
    if rc != 0:
        raise ValueError("Order (%d) and dimension (%d) do not match." % (order, dim // 2 + 1))
    return out


def mfcc_cfg(**kw):
    c = MfccCfg()
    lib().orc_mfcc_default_cfg(C.byref(c))
    keep = []
    for k, v in kw.items():
        if k == "lda":
            if v is not None:
                a = _f32(v); keep.append(a); c.lda = a.ctypes.data
        else:
            setattr(c, k, v)
    c._keep = keep
    return c


def mfcc_chain(x, cfg=None, stage=0):
    cfg = cfg or mfcc_cfg()
    x = _f32(x)
    T = lib().orc_sample_num_blocks(len(x), cfg.blockLen, cfg.shiftLen, cfg.padZeros)
    W = (2 * cfg.delta + 1) * cfg.ncep
    dim = {0: (cfg.outDim if cfg.lda else W), 1: cfg.ncep, 2: cfg.ncep, 3: cfg.filterN, 4: cfg.powN}[stage]
    out = np.zeros((max(T, 1), max(dim, W)), np.float32).reshape(-1)
    n = lib().orc_mfcc_chain(C.byref(cfg), _p(x), len(x), stage, _p(out))
    return out[: n * dim].reshape(n, dim).copy()


def sample_blocks(x, blockLen, shiftLen, padZeros):
    x = _f32(x)
    T = lib().orc_sample_num_blocks(len(x), blockLen, shiftLen, int(padZeros))
    out = np.zeros((T, blockLen), np.float32)
    lib().orc_sample_blocks(_p(x), len(x), blockLen, shiftLen, int(padZeros), _p(out))
    return out


def blockconv(blocks, blockLen, shiftLen):
    b = _f32(blocks); nIn, inLen = b.shape
    T = lib().orc_blockconv_num_frames(nIn, inLen, blockLen, shiftLen)
    out = np.zeros((T, blockLen), np.float32)
    lib().orc_blockconv(_p(b), nIn, inLen, blockLen, shiftLen, _p(out))
    return out


def cosine_matrix(ncep, nmel, typ):
    m = np.zeros((ncep, nmel), np.float32)
    lib().orc_cosine_matrix(ncep, nmel, typ, _p(m))
    return m


def melbank(powN, rate, low, up, filterN, version):
    """-> list of (offset, coeffs float32) rows (feature.cc:1954-2090)."""
    class MB(C.Structure):
        _fields_ = [("filterN", c_int), ("n", c_int), ("offset", C.POINTER(c_int)),
                    ("coefN", C.POINTER(c_int)), ("data", C.POINTER(C.POINTER(c_flt)))]
    h = lib().orc_melbank_create(powN, rate, low, up, filterN, version)
    mb = C.cast(h, C.POINTER(MB)).contents
    rows = [(mb.offset[i], np.array([mb.data[i][j] for j in range(mb.coefN[i])], np.float32)) for i in range(filterN)]
    lib().orc_melbank_free(h)
    return rows


def vtln(pw, ratio, edge, version):
    pw = _f64(pw); T, N = pw.shape
    out = np.zeros_like(pw)
    lib().orc_vtln(_p(pw), T, N, ratio, edge, version, _p(out))
    return out


def cmn_batch(x, devNormFactor=0.0, weights=None):
    x = _f32(x); T, N = x.shape
    out = np.zeros_like(x); mean = np.zeros(N, np.float32); var = np.zeros(N, np.float32)
    w = None if weights is None else _f32(weights)
    L = lib(); L.orc_cmn_batch_w.argtypes = [c_vp, c_int, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp]
    L.orc_cmn_batch_w(_p(x), T, N, devNormFactor, _p(w) if w is not None else None, _p(out), _p(mean), _p(var))
    return out, mean, var


def cmn_runon(x, devNormFactor=0.0, weights=None):
    x = _f32(x); T, N = x.shape
    out = np.zeros_like(x)
    w = None if weights is None else _f32(weights)
    L = lib(); L.orc_cmn_runon_w.argtypes = [c_vp, c_int, c_int, c_dbl, c_vp, c_vp]
    L.orc_cmn_runon_w(_p(x), T, N, devNormFactor, _p(w) if w is not None else None, _p(out))
    return out


def adjacent(x, delta):
    x = _f32(x); T, N = x.shape
    out = np.zeros((T, (2 * delta + 1) * N), np.float32)
    n = lib().orc_adjacent(_p(x), T, N, delta, _p(out))
    return out[:n]


def sgemv_rows(A, X):
    A = _f32(A); X = _f32(X)
    Y = np.zeros((X.shape[0], A.shape[0]), np.float32)
    lib().orc_sgemv_rows(_p(A), A.shape[0], A.shape[1], _p(X), X.shape[0], _p(Y))
    return Y


# ------------------------------------------------------------------ GMM
class Codebooks:
    """Flat codebook set: K codebooks, refN[k] Gaussians each (codebookBasic.h:41 -> refN<=256)."""

    def __init__(self, refN, mean, ivar, det, count=None, scale=None):
        self.refN = np.ascontiguousarray(refN, np.int32)
        self.K = len(self.refN)
        self.off = np.zeros(self.K + 1, np.int32); self.off[1:] = np.cumsum(self.refN)
        self.mean = _f32(mean); self.ivar = _f32(ivar); self.det = _f32(det)
        self.G, self.dimN = self.mean.shape
        self.count = _f32(count if count is not None else np.ones(self.G))
        # float _pi = log(2 pi) * dimN (codebookBasic.cc:170)
        self.pi = np.full(self.K, np.float32(np.log(2.0 * np.pi) * self.dimN), np.float32)
        self.scale = _f32(scale if scale is not None else np.ones(self.K))

    def cstruct(self):
        s = CbSet(self.K, self.dimN, self.refN.ctypes.data, self.off.ctypes.data, self.mean.ctypes.data,
                  self.ivar.ctypes.data, self.det.ctypes.data, self.count.ctypes.data,
                  self.pi.ctypes.data, self.scale.ctypes.data)
        return s


def gmm_score_opt(cb, val, x, native=False):
    x = _f32(x); val = _f32(val); T = x.shape[0]
    score = np.zeros((T, cb.K), np.float32); arg = np.zeros((T, cb.K), np.int32)
    s = cb.cstruct()
    lib(native).orc_gmm_score_opt(C.byref(s), _p(val), _p(x), T, _p(score), _p(arg))
    return score, arg


def gmm_log_lhood(cb, val, x):
    """CodebookBasic::logLhood (codebookBasic.cc:557-609); val None: the reference's val == NULL"""
    x = _f32(x); T = x.shape[0]
    score = np.zeros((T, cb.K), np.float32); arg = np.zeros((T, cb.K), np.int32)
    s = cb.cstruct(); v = _f32(val) if val is not None else None
    L = lib(); L.orc_gmm_log_lhood.argtypes = [c_vp, c_vp, c_vp, c_int, c_vp, c_vp]
    L.orc_gmm_log_lhood(C.byref(s), _p(v) if v is not None else None, _p(x), T, _p(score), _p(arg))
    return score, arg


def gmm_score_all(cb, val, x):
    x = _f32(x); val = _f32(val); T = x.shape[0]
    score = np.zeros((T, cb.K), np.float32)
    s = cb.cstruct()
    lib().orc_gmm_score_all(C.byref(s), _p(val), _p(x), T, _p(score))
    return score


def cbset_save(cb, names, path):
    arr = (C.c_char_p * cb.K)(*[n.encode() for n in names])
    s = cb.cstruct()
    return lib().orc_cbset_save(C.byref(s), arr, path.encode())


def cbset_load(path):
    names = C.POINTER(C.c_char_p)()
    p = lib().orc_cbset_load(path.encode(), C.byref(names))
    if not p:
        return None, None
    s = p.contents
    K, D = s.K, s.dimN
    refN = np.ctypeslib.as_array(C.cast(s.refN, C.POINTER(c_int)), (K,)).copy()
    G = int(refN.sum())
    g = lambda ptr, shp: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(c_flt)), shp).copy()
    cb = Codebooks(refN, g(s.mean, (G, D)), g(s.ivar, (G, D)), g(s.det, (G,)), g(s.count, (G,)))
    nm = [names[k].decode() for k in range(K)]
    return cb, nm


# ------------------------------------------------------------------ WFST + decoder
class Wfst:
    def __init__(self):
        self.h = lib().orc_wfst_new()

    def __del__(self):
        try:
            lib().orc_wfst_free(self.h)
        except Exception:
            pass

    def add_arc(self, s1, s2, i, o, cost=0.0):
        return lib().orc_wfst_add_arc(self.h, s1, s2, i, o, cost)

    def add_final(self, s, cost=0.0):
        return lib().orc_wfst_add_final(self.h, s, cost)

    def read(self, path, binary=False):
        return lib().orc_wfst_read(self.h, path.encode(), int(binary))

    def read_dynamic(self, path, noSelfLoops=False):
        L = lib(); L.orc_wfst_read_dynamic.argtypes = [c_vp, C.c_char_p, c_int]
        return L.orc_wfst_read_dynamic(self.h, path.encode(), int(noSelfLoops))

    def write(self, path, binary=True):
        return lib().orc_wfst_write(self.h, path.encode(), int(binary))

    def export(self):
        L = lib(); n = L.orc_wfst_num_nodes(self.h); a = L.orc_wfst_num_arcs(self.h)
        d = dict(nodeState=np.zeros(n, np.uint32), nodeFinal=np.zeros(n, np.int32), nodeCost=np.zeros(n, np.float32),
                 arcOff=np.zeros(n + 1, np.int32), arcDst=np.zeros(max(a, 1), np.int32), arcIn=np.zeros(max(a, 1), np.uint32),
                 arcOut=np.zeros(max(a, 1), np.uint32), arcCost=np.zeros(max(a, 1), np.float32))
        L.orc_wfst_export(self.h, *[_p(d[k]) for k in ("nodeState", "nodeFinal", "nodeCost", "arcOff", "arcDst", "arcIn", "arcOut", "arcCost")])
        for k in ("arcDst", "arcIn", "arcOut", "arcCost"):
            d[k] = d[k][:a]
        return d

    def decode(self, scores, beam=100.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silenceX=0xFFFFFFFF, dump=False, lattice=False, eosX=0,
               latticeFile=None, writeData=True, topN=0, gmmRows=False):
        """lattice=True: also _Decoder::lattice() (decoder.h:805-953) -> out["lattice"] = dict(nodeFinal, from, to, in, out, start, end, ac, lm), edges
        in creation order; latticeFile: Lattice::write(file, useSymbols=False, writeData) (lattice.cc:715-757)"""
        sc = _f32(scores); T, nDist = sc.shape
        cfg = DecCfg(beam, lmScale, lmPenalty, silPenalty, silenceX, int(dump), int(topN))
        res = DecResult(); lat = Lattice()
        class GmmRows(C.Structure):
            _fields_ = [("n", c_int), ("inX", C.POINTER(C.c_uint)), ("startX", C.POINTER(c_int)), ("endX", C.POINTER(c_int)), ("score", C.POINTER(C.c_double))]
        rows = GmmRows()
        L = lib(); L.orc_decode_ex.argtypes = [c_vp, C.POINTER(DecCfg), c_vp, c_int, c_int, C.POINTER(DecResult), C.POINTER(Lattice), C.c_uint, C.POINTER(GmmRows)]
        rc = L.orc_decode_ex(self.h, C.byref(cfg), _p(sc), T, nDist, C.byref(res), C.byref(lat) if lattice else None, int(eosX), C.byref(rows) if gmmRows else None)
        if rc != 0:
            return dict(rc=rc)
        gmm = None
        if gmmRows:
            n = rows.n
            gmm = None if n < 0 else [(int(rows.inX[i]), int(rows.startX[i]), int(rows.endX[i]), float(rows.score[i])) for i in range(n)]
            L.orc_gmm_rows_free.argtypes = [C.POINTER(GmmRows)]; L.orc_gmm_rows_free(C.byref(rows))
        latd = None
        if lattice:
            n, e = lat.nNodes, lat.nEdges
            g = lambda ptr, dt, k: np.array(ptr[:k], dt) if k > 0 else np.zeros(0, dt)
            latd = {"nodeFinal": g(lat.nodeFinal, np.int32, n), "from": g(lat.from_, np.int32, e), "to": g(lat.to, np.int32, e), "in": g(lat.in_, np.uint32, e),
                    "out": g(lat.out, np.uint32, e), "start": g(lat.start, np.int32, e), "end": g(lat.end, np.int32, e), "ac": g(lat.ac, np.float64, e),
                    "lm": g(lat.lm, np.float64, e)}
            if latticeFile is not None:
                L.orc_lattice_write.argtypes = [C.POINTER(Lattice), C.c_char_p, c_int]
                rcw = L.orc_lattice_write(C.byref(lat), latticeFile.encode(), int(writeData))
                if rcw != 0:
                    raise ValueError("lattice write failed (%d)" % rcw)
            L.orc_lattice_free.argtypes = [C.POINTER(Lattice)]; L.orc_lattice_free(C.byref(lat))
        out = dict(rc=0, score=res.score, ac=res.ac, lm=res.lm, frames=res.frames, reachedFinal=bool(res.reachedFinal),
                   arcs=np.array(res.arcs[:res.nArcs], np.int32), arcFrames=np.array(res.arcFrames[:res.nArcs], np.int32),
                   words=np.array(res.words[:res.nWords], np.uint32), activeHypos=res.activeHypos,
                   activeCount=np.array(res.activeCount[:res.nActive], np.int32),
                   topScore=np.array(res.topScore[:res.nActive], np.float64), finalStatesN=res.finalStatesN)
        if latd is not None:
            out["lattice"] = latd
        if gmmRows:
            out["gmmRows"] = gmm          # _Decoder::writeGMM's label runs (decoder.h:1018-1102), last run first: (input symbol, first frame, last frame, score)
        if dump:
            n = res.nActive
            off = np.array(res.dumpOff[: n + 1], np.int64)
            N = int(off[-1])
            out["dumpOff"] = off
            out["dumpNode"] = np.array(res.dumpNode[:N], np.int32); out["dumpArc"] = np.array(res.dumpArc[:N], np.int32)
            out["dumpAc"] = np.array(res.dumpAc[:N], np.float32); out["dumpLm"] = np.array(res.dumpLm[:N], np.float32)
        lib().orc_dec_result_free(C.byref(res))
        return out
