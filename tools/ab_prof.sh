#!/bin/bash
# timing + in-kernel phase profile of library variants, no result check (for experiments that change results on purpose)
cd $GRAFT_REPO_ROOT
P=distantspeechrecognition-mirror_amd/lib
ARGS="--utts 1024 --frames 100 --reps 3 --beam 53.79"
cp $P/libdsr_hip.so $P/keep.so
for v in "$@"; do
  cp $P/var/$v/libdsr_hip.so $P/libdsr_hip.so
  timeout -k 10 300 python tools/bench_viterbi.py $ARGS > gpurun_out/ab_$v.log 2>&1
  DSR_VITERBI_PROF=1 timeout -k 10 300 python tools/bench_viterbi.py $ARGS --reps 1 > gpurun_out/ab_${v}_prof.log 2>&1
  echo "$v: $(grep -E 'streams=' gpurun_out/ab_$v.log | cut -d: -f2)"; grep prof gpurun_out/ab_${v}_prof.log | tail -n 1
done
cp $P/keep.so $P/libdsr_hip.so
