#!/bin/bash
# end-of-utterance cost of decoder builds (lib/var/<name>): 256 utterances x 1000 frames, profiling ticks p8 (end expansion) and p9 (best token, traceback, output)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export DSR_LIB_VARIANT=$v                                       # (dsr/_capi.py loads lib/var/$v: the shipped library is never touched)
  echo "$v: $(timeout -k 10 300 python tools/bench_viterbi.py --utts 256 --frames 1000 --reps 2 --beam 53.79 2>&1 | grep 'streams=' | cut -d: -f2 | cut -d, -f1)"
done
