#!/bin/bash
# batches between one and two utterances per workgroup: run to completion takes two rounds, time-sliced the batch takes its share
cd $GRAFT_REPO_ROOT
for u in 320 384 512 768; do for sg in 0 125; do
  echo "utts $u DSR_VITERBI_SEG=$sg: $(DSR_VITERBI_SEG=$sg timeout -k 10 400 python tools/bench_viterbi.py --utts $u --frames 1000 --reps 2 --beam 53.79 2>&1 | grep -E 'streams=' | cut -d: -f2 | cut -d, -f1)"
done; done
