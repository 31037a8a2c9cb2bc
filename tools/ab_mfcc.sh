#!/bin/bash
# the MFCC frame kernel with stages compiled out (OBJ=k_mfcc tools/build_variants.sh name "-DDSR_MFCC_NOFFT" / "-DDSR_MFCC_NOTAIL"): where a frame's time goes
cd $GRAFT_REPO_ROOT
for v in "$@"; do echo "$v: $(DSR_LIB_VARIANT=$v python tools/bench_mfcc.py --utts 1000 --stage 1 2>&1 | tail -1 | cut -c1-110)"; done
