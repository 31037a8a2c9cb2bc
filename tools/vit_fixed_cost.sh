#!/bin/bash
# per-phase ticks of the decoder at the tuned beam and at a beam that leaves a few dozen placements a frame: what a frame costs whatever the work
cd $GRAFT_REPO_ROOT
for b in 53.79 30 12; do
  echo "== beam $b"
  DSR_VITERBI_PROF=1 timeout -k 10 300 python tools/bench_viterbi.py --utts 256 --frames 400 --reps 1 --beam $b 2>&1 | grep -E "streams=|prof"
done
