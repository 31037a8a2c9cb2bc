#!/bin/bash
# the software-pipelined GMM shape (DSR_GMM_SP=1) against the two-waves-per-SIMD shape: parity tests, micro benchmark at the pipe's shape, the pipe's stage time
cd $GRAFT_REPO_ROOT
for sp in 0 1; do
  export DSR_GMM_SP=$sp
  echo "== DSR_GMM_SP=$sp"
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "gmm" 2>&1 | tail -3 || exit 1
  timeout -k 10 200 python tools/bench_gmm.py --frames 1005600 --K 1024 --R 4 --reps 5 2>&1 | tail -3 || exit 1
  timeout -k 10 300 python bench.py --serial --no-cpu --no-verify --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('pipe gmm', d['stages']['gmm']['ms'], 'step', round(d['ms_per_step'],1))" || exit 1
done
