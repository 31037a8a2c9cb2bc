#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "gmm" 2>&1 | tail -3
for st in 1 0; do for d in 0 4; do echo "stagger $st spdbg $d: $(DSR_GMM_STAGGER=$st DSR_GMM_SPDBG=$d python tools/bench_gmm.py --frames 1005600 --K 1024 --R 4 --modes 2 --reps 5 2>&1 | grep 'mode 2:')"; done; done
