#!/bin/bash
# the pipe's GMM stage over library variants (lib/var/<name>, tools/build_gmm_variants.sh): bench.py --serial stage table on one box
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export DSR_LIB_VARIANT=$v                                       # (dsr/_capi.py loads lib/var/$v: the shipped library is never touched)
  echo "$v: $(python bench.py --serial --no-cpu --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('gmm', d['stages']['gmm']['ms'], 'step', round(d['ms_per_step'],1))")"
done
