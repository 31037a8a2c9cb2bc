#!/bin/bash
# BASELINE configs[4] (64 channels, long-form streams in ten-second blocks): the bench line and the rocprofv3 kernel stats of the same command.  Output under gpurun_out/r03/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R && python3 bench.py --workload config5 > $O/config5.json 2> $O/config5.err || { tail -5 $O/config5.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace5 -o c5 -- python3 $R/bench.py --workload config5 --no-cpu > $O/config5_traced.json 2> $O/config5_traced.err || exit 1
cp $(find $O/trace5 -name "*kernel_stats.csv" | head -n 1) $O/config5_kernel_stats.csv
rm -rf $O/trace5
tail -c 1500 $O/config5.json
