#!/bin/bash
# GMM micro benchmark over the library variants in lib/var/<name>/ (tools/build_gmm_variants.sh) on one box
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export DSR_LIB_VARIANT=$v                                       # (dsr/_capi.py loads lib/var/$v: the shipped library is never touched)
  for d in 0 6; do echo "$v dbg $d: $(DSR_GMM_DBG=$d python tools/bench_gmm.py --K 1024 --R 4 --modes 2 --no-argmin --reps 5 2>&1 | grep 'mode 2')"; done
done
