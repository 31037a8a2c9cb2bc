#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 53.79 40 30 12; do
  echo "== beam $b: $(timeout -k 10 300 python tools/bench_viterbi.py --utts 256 --frames 400 --reps 2 --beam $b 2>&1 | grep -E 'streams=')"
done
