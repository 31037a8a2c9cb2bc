#!/bin/bash
# I-cache / issue counters of the decoder micro benchmark (separate passes; kernel trace only)
cd /tmp && export TMPDIR=/tmp
true
true
echo
cd $GRAFT_REPO_ROOT
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  tag=v_$(echo $set | cut -d' ' -f1)
  bash tools/pmc.sh $tag "$set" $GRAFT_REPO_ROOT/tools/bench_viterbi.py --utts 1024 --frames 100 --reps 1 --beam 53.79 | grep viterbi
done
