"""asr.gaussian: CodebookSetBasicPtr / DistribSetBasicPtr (gaussian.i:281-283,465-467) for 1:1 distribution/codebook models, over dsr_gmm_* and
dsr_distribset_* (include/dsr.h sections 4 and 8)."""
import ctypes as C

import numpy as np

from .. import _capi as K


def _desc(path):
    rows = []
    for line in open(path):
        if line[:1] == ";" or not line.split():
            continue
        rows.append(line.split())
    return rows


class CodebookSetBasicPtr(object):
    """descFile lines: name featureName refN dimN covType (codebookBasic.cc:804-828); cbkFile: big-endian set file."""

    def __init__(self, descFile="", fs=None, cbkFile=""):
        self._desc = _desc(descFile) if descFile else []
        self._fs, self._cbkFile = fs, cbkFile
        self.names = [r[0] for r in self._desc]
        self.featureName = self._desc[0][1] if self._desc else None

    def ncbks(self):
        return len(self._desc)

    def feature(self):
        return self._fs.feature(self.featureName)


class DistribBasicPtr(object):
    """Distrib::score(frameX) (distribBasic.h:48-50) of one distribution of a set"""

    def __init__(self, dss, distX):
        self._dss, self._x = dss, distX

    def name(self):
        return K.load().dsr_distribset_name(self._dss._ds, self._x).decode()

    def score(self, frameX):
        s = C.c_float(); st = K.load().dsr_distribset_score(self._dss._ds, self._x, int(frameX), C.byref(s))
        if st == K.E_ITERATOR:
            raise StopIteration
        K.check(st); return s.value


class DistribSetBasicPtr(object):
    """descFile lines: name codebookName (distribBasic.cc:218-232); distFile: big-endian set file."""

    def __init__(self, cbs, descFile="", distFile="", gmmMode=0):
        self._cbs = cbs; self._desc = _desc(descFile) if descFile else []
        self.names = [r[0] for r in self._desc]
        self.gmm = K.Gmm(files=(cbs._cbkFile, distFile))
        if self._desc and len(self._desc) != self.gmm.K:
            raise K.DsrError(4, "%d distributions described, %d in the file" % (len(self._desc), self.gmm.K))
        self._feat = cbs.feature(); self._ds = C.c_void_p()
        K.check(K.load().dsr_distribset_create(self.gmm.h, self._feat._h, int(gmmMode), C.byref(self._ds)))

    def __del__(self):
        try:
            if getattr(self, "_ds", None):
                K.load().dsr_distribset_destroy(self._ds)
        except Exception:
            pass

    def ndists(self):
        return self.gmm.K

    def index(self, name):
        x = C.c_int(); K.check(K.load().dsr_distribset_find(self._ds, name.encode(), C.byref(x))); return x.value

    def find(self, key):
        """find(name) / find(dsX) (distribBasic.h:183-190)"""
        x = self.index(key) if isinstance(key, str) else int(key)
        if not 0 <= x < self.gmm.K:
            raise K.DsrError(6, "distribution %d of %d" % (x, self.gmm.K))
        return DistribBasicPtr(self, x)

    def resetCache(self):
        K.check(K.load().dsr_distribset_reset_cache(self._ds))

    def resetFeature(self):
        K.check(K.load().dsr_distribset_reset_feature(self._ds))

    def score_all_frames(self, feats, mode=0):
        """feats: cuda float32 [T][dimN] -> costs [T][ndists] (row t = Distrib::score(t) of every distribution)."""
        return self.gmm.score(feats, mode=mode, want_argmin=False)[0]
