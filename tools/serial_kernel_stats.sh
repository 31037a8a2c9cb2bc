#!/bin/bash
# kernel times of the pipe with the steps one after the other (no overlap between steps: a kernel's duration is its own)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/serial_stats; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 $R/bench.py --serial --steps 4 --warmup 2 --no-cpu --no-verify > $O/bench.json 2> $O/bench.err || exit 1
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "dsr::" in r["Name"]: print("%-62s %4s calls  avg %9.1f us  min %9.1f" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
