"""asr.gaussian: CodebookSetBasicPtr / DistribSetBasicPtr (gaussian.i:281-283,465-467) for 1:1 distribution/codebook models."""
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


class DistribSetBasicPtr(object):
    """descFile lines: name codebookName (distribBasic.cc:218-232); distFile: big-endian set file."""

    def __init__(self, cbs, descFile="", distFile=""):
        self._cbs = cbs; self._desc = _desc(descFile) if descFile else []
        self.names = [r[0] for r in self._desc]
        self.gmm = K.Gmm(files=(cbs._cbkFile, distFile))
        if self._desc and len(self._desc) != self.gmm.K:
            raise K.DsrError(4, "%d distributions described, %d in the file" % (len(self._desc), self.gmm.K))

    def ndists(self):
        return self.gmm.K

    def score_all_frames(self, feats, mode=0):
        """feats: cuda float32 [T][dimN] -> costs [T][ndists] (row t = Distrib::score(t) of every distribution)."""
        return self.gmm.score(feats, mode=mode, want_argmin=False)[0]
