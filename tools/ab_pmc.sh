#!/bin/bash
# counter pass of the decoder micro benchmark for library variants: tools/ab_pmc.sh "<COUNTERS>" v1 v2 ...
cd $GRAFT_REPO_ROOT
ctrs=$1; shift
for v in "$@"; do
  export DSR_LIB_VARIANT=$v
  echo "== $v"; bash tools/pmc.sh ab_$v "$ctrs" $GRAFT_REPO_ROOT/tools/bench_viterbi.py --utts 1024 --frames 100 --reps 1 --beam 53.79 | grep k_viterbi
done
