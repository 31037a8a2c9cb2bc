#!/bin/bash
# the one-wave-per-SIMD MFMA shape against the two-wave shape for codebooks of 4 .. 32 Gaussians (G = 4096, 1 M frames)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "gmm" 2>&1 | tail -3
for r in 4 8 16 32; do for sp in 1 0; do
  echo "R=$r DSR_GMM_SP=$sp: $(DSR_GMM_SP=$sp timeout -k 10 200 python tools/bench_gmm.py --frames 1005600 --K $((4096 / r)) --R $r --reps 5 --modes 0,2 2>&1 | grep -E 'mode 2' | tr '\n' ' ')"
done; done
