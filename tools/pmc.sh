#!/bin/bash
# usage: tools/pmc.sh <tag> "<COUNTER COUNTER ...>" <python script and args...>
# one rocprofv3 counter pass (kernel trace only, as the pool requires); summary of the named kernels to gpurun_out/<tag>.txt
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc $ctrs -d $out -o run --output-format csv -- python3 "$@" > $out/stdout.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: "%.4g" % v for c, v in acc[k].items()}, "launches", max(cnt[(k, c)] for c in acc[k]))
PY
