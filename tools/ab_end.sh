#!/bin/bash
# end-of-utterance cost of decoder builds (lib/var/<name>): 256 utterances x 1000 frames, profiling ticks p8 (end expansion) and p9 (best token, traceback, output)
cd $GRAFT_REPO_ROOT
P=distantspeechrecognition-mirror_amd/lib
cp $P/libdsr_hip.so $P/keep.so
for v in "$@"; do
  cp $P/var/$v/libdsr_hip.so $P/libdsr_hip.so
  echo "$v: $(timeout -k 10 300 python tools/bench_viterbi.py --utts 256 --frames 1000 --reps 2 --beam 53.79 2>&1 | grep 'streams=' | cut -d: -f2 | cut -d, -f1)"
done
cp $P/keep.so $P/libdsr_hip.so
