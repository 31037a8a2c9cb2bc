#!/bin/bash
# timing + in-kernel phase profile of library variants, no result check (for experiments that change results on purpose)
cd $GRAFT_REPO_ROOT
ARGS="--utts 1024 --frames 100 --reps 3 --beam 53.79"
for v in "$@"; do
  export DSR_LIB_VARIANT=$v                                       # (dsr/_capi.py loads lib/var/$v: the shipped library is never touched)
  timeout -k 10 300 python tools/bench_viterbi.py $ARGS > gpurun_out/ab_$v.log 2>&1
  DSR_VITERBI_PROF=1 timeout -k 10 300 python tools/bench_viterbi.py $ARGS --reps 1 > gpurun_out/ab_${v}_prof.log 2>&1
  echo "$v: $(grep -E 'streams=' gpurun_out/ab_$v.log | cut -d: -f2)"; grep prof gpurun_out/ab_${v}_prof.log | tail -n 1
done
