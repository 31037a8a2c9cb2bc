#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 6 7; do echo "spdbg $d: $(DSR_GMM_SPDBG=$d python tools/bench_gmm.py --frames 1005600 --K 1024 --R 4 --modes 2 --reps 5 2>&1 | grep 'mode 2:')"; done
