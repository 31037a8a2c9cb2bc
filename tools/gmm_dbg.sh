set -e
for d in 0 1 2 4 6 8 12; do echo "== DBG $d"; DSR_GMM_DBG=$d python tools/bench_gmm.py --K 1024 --R 4 --modes 2 --no-argmin; done
echo "== with argmin"; python tools/bench_gmm.py --K 1024 --R 4 --modes 2
