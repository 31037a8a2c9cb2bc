#!/bin/bash
cd $GRAFT_REPO_ROOT
for sg in 0 125; do
  echo "== DSR_VITERBI_SEG=$sg"; DSR_VITERBI_PROF=1 DSR_VITERBI_SEG=$sg timeout -k 10 400 python tools/bench_viterbi.py --utts 1000 --frames 1000 --reps 1 --beam 53.79 2>&1 | grep -E 'streams=|prof' | tail -3
done
