#!/bin/bash
# time slicing of the decoder (DSR_VITERBI_SEG frames per segment, 0 = every utterance run to completion): micro benchmark + result check against the memory path
cd $GRAFT_REPO_ROOT
for sg in 0 125 250 60; do
  echo "== DSR_VITERBI_SEG=$sg: $(DSR_VITERBI_SEG=$sg timeout -k 10 400 python tools/bench_viterbi.py --utts 1000 --frames 1000 --reps 2 --beam 53.79 ${CHECK} 2>&1 | grep -E 'streams=|differ|Error|error' | cut -c1-200 | tr '\n' ' ')"
done
