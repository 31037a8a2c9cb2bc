#!/bin/bash
# kernel times of the MFMA scoring path at the pipe's shape (and, with "pmc", the matrix-pipe counters of the scoring kernel)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for sp in ${SPS:-1}; do
  export DSR_GMM_SP=$sp
  (cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/gsp$sp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gsp$sp -o run --output-format csv -- python3 $R/tools/bench_gmm.py --frames 1005600 --K 1024 --R 4 --modes 2 --reps 5 > /dev/null 2>&1)
  echo "== SP=$sp"; python3 - $R/gpurun_out/gsp$sp <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gmm" in r["Name"]: print(r["Name"][:50], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
  if [ "$1" = "pmc" ]; then
  bash tools/pmc.sh gsp${sp}a "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES" $R/tools/bench_gmm.py --frames 1005600 --K 1024 --R 4 --modes 2 --reps 2 | grep -i "gmm_mfma"
  bash tools/pmc.sh gsp${sp}b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" $R/tools/bench_gmm.py --frames 1005600 --K 1024 --R 4 --modes 2 --reps 2 | grep -i "gmm_mfma"
  fi
done
