#!/bin/bash
# Decoder micro benchmark over the library variants in lib/var/<name>/ (tools/build_variants.sh): same box, same inputs, two rounds.
# The variant is selected through DSR_LIB_VARIANT (dsr/_capi.py); the shipped library is never touched.
cd $GRAFT_REPO_ROOT
ARGS="--utts 1024 --frames 100 --reps 3 --beam 53.79"
for round in 1 2; do
for v in "$@"; do
  export DSR_LIB_VARIANT=$v
  if [ $round = 1 ]; then timeout -k 10 300 python tools/bench_viterbi.py $ARGS --check > gpurun_out/ab_$v.log 2>&1 || { echo "$v FAILED"; tail -n 5 gpurun_out/ab_$v.log; exit 1; }
  else DSR_VITERBI_PROF=1 timeout -k 10 300 python tools/bench_viterbi.py $ARGS --reps 1 > gpurun_out/ab_${v}_prof.log 2>&1; timeout -k 10 300 python tools/bench_viterbi.py $ARGS > gpurun_out/ab_$v.log 2>&1 || exit 1; fi
  echo "$v r$round: $(grep -E 'streams=' gpurun_out/ab_$v.log | cut -d: -f2 | cut -d, -f1) $(grep -E 'differ' gpurun_out/ab_$v.log | cut -d: -f2)"
  if [ $round = 2 ]; then grep prof gpurun_out/ab_${v}_prof.log | tail -n 1; fi
done; done
