"""asr.dictionary.LexiconPtr (asr/dictionary/distribTree.h:40-65, distribTree.cc:46-87): symbol <-> index by line order."""


class LexiconPtr(object):
    def __init__(self, nm="Lexicon", fileName=""):
        self._nm, self._syms, self._idx = nm, [], {}
        if fileName:
            self.read(fileName)

    def read(self, fileName):
        self._syms, self._idx = [], {}
        for line in open(fileName):
            if line[:1] == ";":
                continue
            tok = line.split()
            if not tok:
                continue
            if tok[0] in self._idx:              # "Symbol %s already exists." -- kept, not an error
                continue
            self._idx[tok[0]] = len(self._syms); self._syms.append(tok[0])   # the index column is ignored

    def index(self, symbol):
        if symbol not in self._idx:
            from .. import _capi as K
            raise K.DsrError(11, "Could not find key %s" % symbol)     # jkey_error (mlist.h:109-114)
        return self._idx[symbol]

    def symbol(self, index):
        return self._syms[index]

    def size(self):
        return len(self._syms)

    def isPresent(self, symbol):
        return symbol in self._idx
