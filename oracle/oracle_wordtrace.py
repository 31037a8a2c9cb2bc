"""oracle/oracle_wordtrace.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ may import it).

CPU restatement of DecoderWordTrace AS SHIPPED (asr/decoder/decoder.h:1146-1304, asr/decoder/decoder.cc:126-470) on an object graph shaped like the
reference's: Token objects with float scores and a `worse` chain, WordTrace objects, a token list with the reference's insertion order
(_TokenList::insert prepends, replace keeps the holder's place: decoder.h:224-262), over oracle_wfst.FlyWeightSortedOutput.

    _processFirstFrame / _processFrame   decoder.cc:136-183   (beam only: this class's frame loop has no topN branch; epsilon / validEndN kept)
    _expandToEnd                         :185-201             (note the FLOAT lmScoreEdge)
    _notPresent                          :203-211             (compares _uniqueIndices[0] only -- kept)
    _placeOnList                         :213-267             (generateLattice: merge-sort-unique over the worse chains; dereferences wordTrace())
    _hashWordSequence                    :327-349
    _advanceTokens                       :357-392
    _expandNode / _expandNodeToEnd       :394-470
    _minorTrace                          :274-325             (lattice links through the word traces), _majorTrace / lattice(): decoder.h:805-953
    decode / _bestToken / bestHypo       decoder.h:639-773

Scores are kept the way the reference keeps them: doubles in flight, rounded to float when a Token is constructed (lattice.h:37-79), score() is the FLOAT
sum of the two floats.  With generateLattice a token that has not yet crossed a word boundary has a null word trace and _placeOnList dereferences it
(:239): undefined behaviour in the reference, raised here as NullWordTrace with the frame and state where it happens.

Parity unpinned: no driver, test or output of the reference uses this class; restated from the source text.
"""
import struct

HUGE = float("inf")


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


class NullWordTrace(Exception):
    """_placeOnList dereferenced the word trace of a token that has none (decoder.cc:239)"""


class EndOfSamples(Exception):
    """jiterator_error"""


class WordTrace(object):
    __slots__ = ("wordX", "wordSequenceX", "endX", "tokenList")

    def __init__(self, wordX, wordSequenceX, endX, tokenList=None):
        self.wordX, self.wordSequenceX, self.endX, self.tokenList = wordX, wordSequenceX, endX, tokenList


class Token(object):
    __slots__ = ("ac", "lm", "frameX", "edge", "wordTrace", "worse")

    def __init__(self, acs, lms, frameX, edge, wordTrace=None, worse=None):
        self.ac, self.lm, self.frameX, self.edge, self.wordTrace, self.worse = f32(acs), f32(lms), frameX, edge, wordTrace, worse

    def score(self):
        return f32(self.ac + self.lm)


class TokenList(object):
    """_TokenList (decoder.h:48-320): hash by state + a list that new holders are PREPENDED to; replace() swaps the token in its holder"""

    def __init__(self):
        self.holders = []          # iteration order = reversed insertion order
        self.byState = {}

    def clear(self):
        self.holders = []; self.byState = {}

    def isPresent(self, s):
        return s in self.byState

    def token(self, s):
        return self.byState[s][0]

    def insert(self, s, tok):
        h = [tok]; self.byState[s] = h; self.holders.append(h)

    def replace(self, s, tok):
        self.byState[s][0] = tok

    def __iter__(self):
        for h in reversed(self.holders):
            yield h[0]

    def activeTokens(self):
        return len(self.holders)


class DecoderWordTrace(object):
    def __init__(self, scoreFn, beam=100.0, lmScale=12.0, lmPenalty=0.0, silPenalty=0.0, silenceX=0xFFFFFFFF, eosX=0, epsilon=0.0, validEndN=30,
                 generateLattice=True, propagateN=5, fastHash=False, insertSilence=False, outputLexicon=None):
        """scoreFn(distX, frameX) -> cost or raises EndOfSamples (the _dist->find(distX - 1)->score(_frameX) of decoder.cc:403)"""
        self.scoreFn = scoreFn
        self.beam, self.lmScale, self.lmPenalty, self.silPenalty = beam, lmScale, lmPenalty, silPenalty
        self.silenceX, self.eosX, self.epsilon, self.validEndN = silenceX, eosX, epsilon, validEndN
        self.generateLattice, self.propagateN, self.fastHash, self.insertSilence = generateLattice, propagateN, fastHash, insertSilence
        self.outLex = outputLexicon
        self.wordSequenceHash = []; self.uniqueIndices = [0] * max(1, propagateN); self.tokenX = 0
        self.current = TokenList(); self.next = TokenList()
        self.wfst = None

    def set(self, wfst):
        self.wfst = wfst

    # ---- helpers
    def _hashWordSequence(self, thisToken, wordX):
        if self.fastHash:
            seq = 0
            if thisToken is not None and thisToken.wordTrace is not None:
                seq = thisToken.wordTrace.wordSequenceX
            return (337 * seq + wordX + 1) & 0xFFFFFFFF
        word = self.outLex[wordX] if self.outLex is not None else str(wordX)
        if thisToken is None or thisToken.wordTrace is None:
            new = word
        else:
            new = self.wordSequenceHash[thisToken.wordTrace.wordSequenceX] + " " + word
        if new not in self.wordSequenceHash:
            self.wordSequenceHash.append(new)
        return self.wordSequenceHash.index(new)

    def _advanceTokens(self, edgeA, acScoreEdge, lmScoreEdge, topToken):
        if topToken is None:
            if edgeA.input == self.silenceX:
                lmScoreEdge += self.lmScale * self.silPenalty
            return Token(acScoreEdge, lmScoreEdge, self.frameX, edgeA)
        newTop = None; newTok = None; this = topToken
        while this is not None:
            ac = acScoreEdge + this.ac
            lm = lmScoreEdge + this.lm
            if edgeA.input == self.silenceX and this.edge.input != self.silenceX:
                lm += self.lmScale * self.silPenalty
            tok = Token(ac, lm, self.frameX, edgeA, this.wordTrace)
            if newTop is None:
                newTop = tok
            else:
                newTok.worse = tok
            newTok = tok
            this = this.worse
        return newTop

    def _notPresent(self, wordSeqX):
        for i in range(self.tokenX):
            if self.uniqueIndices[0] == wordSeqX:
                return False
        self.uniqueIndices[self.tokenX] = wordSeqX; self.tokenX += 1
        return True

    def _placeOnList(self, edge, tok):
        ttl = tok.score()
        if ttl < self.topScore and edge.input != 0:
            self.topScore = ttl
        stateX = edge.next.index
        if self.next.isPresent(stateX):
            nextToken = self.next.token(stateX)
            if self.generateLattice:
                self.tokenX = 0
                merge = None; sort = None
                while self.tokenX < self.propagateN and (tok is not None or nextToken is not None):
                    if tok is None:
                        best = nextToken; nextToken = nextToken.worse
                    elif nextToken is None:
                        best = tok; tok = tok.worse
                    elif tok.score() < nextToken.score():
                        best = tok; tok = tok.worse
                    else:
                        best = nextToken; nextToken = nextToken.worse
                    if best.wordTrace is None:
                        raise NullWordTrace("frame %d, state %d" % (self.frameX, stateX))
                    if self._notPresent(best.wordTrace.wordSequenceX):
                        if merge is None:
                            merge = best; sort = merge; sort.worse = None
                        else:
                            sort.worse = best; sort = best; sort.worse = None
                self.next.replace(stateX, merge)
            elif ttl < nextToken.score():
                self.next.replace(stateX, tok)
        else:
            self.next.insert(stateX, tok)

    def _expandNode(self, node, topToken=None):
        acNode = lmNode = 0.0
        if topToken is not None:
            acNode = topToken.ac; lmNode = topToken.lm
        for edge in node.iter_edges():
            distX, wordX = edge.input, edge.output
            acEdge = 0.0 if distX == 0 else self.scoreFn(distX, self.frameX)
            lmEdge = self.lmScale * edge.cost
            newWord = False
            if wordX != 0:
                lmEdge += self.lmScale * self.lmPenalty
                newWord = True
            if edge.next.final:
                ttlNode = acNode + lmNode + acEdge + lmEdge
                if ttlNode < self.topEndScore:
                    self.topEndScore = ttlNode
            wordToken = self._advanceTokens(edge, acEdge, lmEdge, topToken)
            if self.insertSilence and edge.input == self.silenceX and (topToken is None or topToken.edge.input != self.silenceX):
                newWord = True
            if newWord:
                seq = self._hashWordSequence(topToken, wordX)
                wt = WordTrace(wordX, seq, self.frameX, wordToken)
                wordToken = Token(wordToken.ac, wordToken.lm, self.frameX, edge, wt)
            if distX == 0:
                self._expandNode(edge.next, wordToken)
            else:
                self._placeOnList(edge, wordToken)

    def _expandNodeToEnd(self, node, topToken):
        for edge in node.iter_edges():
            if edge.input != 0:
                continue
            wordX = edge.output
            lmEdge = self.lmScale * edge.cost
            if wordX != 0:
                lmEdge += self.lmScale * self.lmPenalty
            wordToken = self._advanceTokens(edge, 0.0, lmEdge, topToken)
            if wordX != 0:
                seq = self._hashWordSequence(topToken, wordX)
                wt = WordTrace(wordX, seq, self.frameX, wordToken)
                wordToken = Token(wordToken.ac, wordToken.lm, self.frameX, edge, wt)
            if edge.next.final:
                lmFinal = f32(self.lmScale * edge.next.cost)             # "float lmScoreFinal"
                endToken = self._advanceTokens(edge, 0.0, lmFinal, wordToken)
                self._placeOnList(edge, endToken)
                if endToken.score() < self.topEndScore:
                    self.topEndScore = endToken.score()
            self._expandNodeToEnd(edge.next, wordToken)

    def _expandToEnd(self):
        self.next.clear()
        for tok in self.current:
            if tok.edge.next.final:
                lmEdge = f32(self.lmScale * tok.edge.next.cost)          # "float lmScoreEdge"
                wordToken = self._advanceTokens(tok.edge, 0.0, lmEdge, tok)
                self._placeOnList(tok.edge, wordToken)
            self._expandNodeToEnd(tok.edge.next, tok)

    # ---- frames
    def _processFirstFrame(self):
        self.topScore = HUGE
        self.frameX = 0; self.validEndX = 0
        self._expandNode(self.wfst.initial)
        self.activeHypos = self.next.activeTokens()

    def _processFrame(self):
        self.current, self.next = self.next, self.current; self.next.clear()
        thresh = self.topScore + self.beam
        self.topScore = self.topEndScore = HUGE
        for tok in self.current:
            if tok.score() > thresh:
                continue
            self._expandNode(tok.edge.next, tok)
        self.activeHypos += self.next.activeTokens()
        if self.epsilon > 0.0 and self.topEndScore > 0.0 and self.topEndScore < self.topScore + self.epsilon:
            self.validEndX += 1
            if self.validEndX == self.validEndN:
                raise EndOfSamples("end of samples!")
        else:
            self.validEndX = 0

    def _bestToken(self):
        best = None; bestScore = HUGE; self.reachedFinal = True
        for tok in self.next:
            if not tok.edge.next.final:
                raise RuntimeError("Node %d is not a final node." % tok.edge.next.index)
            if tok.score() < bestScore:
                bestScore = tok.score(); best = tok
        if best is None:
            self.reachedFinal = False; bestScore = HUGE
            for tok in self.current:
                if tok.score() < bestScore:
                    bestScore = tok.score(); best = tok
        return best

    def decode(self):
        """_Decoder::decode (decoder.h:687-737): returns double(ac) + double(lm) of the best token"""
        self.wordSequenceHash = []; self.current.clear(); self.next.clear(); self.topEndScore = HUGE
        self._processFirstFrame()
        try:
            while True:
                self.frameX += 1
                self._processFrame()
        except EndOfSamples:
            pass
        self.frameX -= 1
        self._expandToEnd()
        tok = self._bestToken()
        self.best = tok
        return tok.ac + tok.lm

    def finalStatesN(self):
        return sum(1 for t in self.next if t.edge.next.final)

    def bestHypoIds(self, useInputSymbols=False):
        """bestHypo (decoder.h:748-773): this class's tokens carry no prev(): the walk ends after the best token's own edge"""
        e = self.best.edge
        x = e.input if useInputSymbols else e.output
        return [x] if x != 0 else []

    def wordTraceIds(self):
        """the words along the best token's word traces, first word first (what _minorTrace / _majorTrace turn into lattice links)"""
        out = []; wt = self.best.wordTrace
        while wt is not None:
            out.append(wt.wordX)
            wt = wt.tokenList.wordTrace if wt.tokenList is not None else None
        return out[::-1]
