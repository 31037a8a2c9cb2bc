#!/bin/bash
# the pipe's GMM stage over library variants (lib/var/<name>, tools/build_gmm_variants.sh): bench.py --serial stage table on one box
cd $GRAFT_REPO_ROOT
P=distantspeechrecognition-mirror_amd/lib
cp $P/libdsr_hip.so $P/keep.so
for v in "$@"; do
  cp $P/var/$v/libdsr_hip.so $P/libdsr_hip.so
  echo "$v: $(python bench.py --serial --no-cpu --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('gmm', d['stages']['gmm']['ms'], 'step', round(d['ms_per_step'],1))")"
done
cp $P/keep.so $P/libdsr_hip.so
