#!/bin/bash
# The round's profile evidence, one gpurun call: (1) rocprofv3 --kernel-trace --stats of the default bench command, (2) separate --pmc passes
# (kernel trace only, as the pool requires) for HBM traffic and the decoder's issue/wait split.  Output under gpurun_out/r03/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --no-cpu --no-verify > $O/bench_traced.json 2> $O/bench_traced.err || exit 1
cp $(find $O/trace -name "*kernel_stats.csv" | head -n 1) $O/bench_kernel_stats.csv
python3 - $O <<'PY'
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_viterbi" in r["Kernel_Name"]]
with open(d + "/bench_viterbi_launches.txt", "w") as o:
    for r in rows:
        o.write("%s grid %s dur_ms %.3f\n" % (r["Kernel_Name"][:20], r.get("Grid_Size_X", r.get("Grid_Size", "?")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $O/trace
cd $R
ARGS="$R/bench.py --serial --steps 1 --warmup 1 --no-cpu --no-verify --beam 53.787"     # one pipe: probe + two full-batch launches of every kernel
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=r03_$(echo $set | cut -d' ' -f1)
  bash tools/pmc.sh $tag "$set" $ARGS > $O/pmc_$tag.txt 2>&1 || echo "pmc pass $tag failed"
  rm -rf $R/gpurun_out/pmc_$tag
done
ls $O
