#!/bin/bash
# counter pass of the decoder micro benchmark for library variants: tools/ab_pmc.sh "<COUNTERS>" v1 v2 ...
cd $GRAFT_REPO_ROOT
P=distantspeechrecognition-mirror_amd/lib; ctrs=$1; shift
cp $P/libdsr_hip.so $P/keep.so
for v in "$@"; do
  cp $P/var/$v/libdsr_hip.so $P/libdsr_hip.so
  echo "== $v"; bash tools/pmc.sh ab_$v "$ctrs" $GRAFT_REPO_ROOT/tools/bench_viterbi.py --utts 1024 --frames 100 --reps 1 --beam 53.79 | grep k_viterbi
done
cp $P/keep.so $P/libdsr_hip.so
