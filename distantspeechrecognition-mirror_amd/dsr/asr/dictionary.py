"""asr.dictionary.LexiconPtr (asr/dictionary/distribTree.h:40-65, distribTree.cc:36-133) over the C-ABI container dsr_lexicon_*:
symbol <-> index by line order of the file (its index column is ignored), ';' comment lines, repeated symbols skipped."""
import ctypes as C

from .. import _capi as K


class LexiconPtr(object):
    def __init__(self, nm="Lexicon", fileName=""):
        L = K.load(); self._h = C.c_void_p()
        K.check(L.dsr_lexicon_create(nm.encode(), (fileName or "").encode(), C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                K.load().dsr_lexicon_destroy(self._h)
        except Exception:
            pass

    def read(self, fileName):
        K.check(K.load().dsr_lexicon_read(self._h, fileName.encode()))

    def write(self, fileName, writeHeader=False):
        K.check(K.load().dsr_lexicon_write(self._h, fileName.encode(), int(writeHeader)))

    def clear(self):
        K.check(K.load().dsr_lexicon_clear(self._h))

    def name(self):
        return K.load().dsr_lexicon_name(self._h).decode()

    def index(self, symbol, create=False):
        i = C.c_uint()
        K.check(K.load().dsr_lexicon_index(self._h, symbol.encode(), int(create), C.byref(i)))      # jkey_error (mlist.h:109-114) when absent
        return i.value

    def symbol(self, index):
        p = C.c_char_p()
        K.check(K.load().dsr_lexicon_symbol(self._h, int(index), C.byref(p)))
        return p.value.decode()

    def size(self):
        return K.load().dsr_lexicon_size(self._h)

    def isPresent(self, symbol):
        return bool(K.load().dsr_lexicon_is_present(self._h, symbol.encode()))

    def __iter__(self):
        return (self.symbol(i) for i in range(self.size()))
