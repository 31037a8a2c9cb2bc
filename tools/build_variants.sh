#!/bin/bash
# usage: [OBJ=k_viterbi] tools/build_variants.sh name1 "-DFLAGS" name2 "-DFLAGS" ...   -> distantspeechrecognition-mirror_amd/lib/var/<name>/libdsr_hip.so
# OBJ: the translation unit the variants rebuild (default k_viterbi); a name of the form name=path/to/source.hip builds that source file instead of csrc/$OBJ.hip.
# Variants are selected at run time through DSR_LIB_VARIANT (dsr/_capi.py); the shipped library is never touched.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/distantspeechrecognition-mirror_amd
OBJ=${OBJ:-k_viterbi}
OTHERS=$(ls lib/*.o | grep -v "/$OBJ.o")
EXACT="-ffp-contract=off"; case $OBJ in k_filterbank|k_beamform) EXACT="";; esac
while [ $# -gt 1 ]; do
  n=$1; fl=$2; shift 2; src=csrc/$OBJ.hip
  case $n in *=*) src=${n#*=}; n=${n%%=*};; esac
  mkdir -p lib/var/$n
  ( /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -I../include -Icsrc $EXACT $fl -Rpass-analysis=kernel-resource-usage -c $src -o lib/var/$n/$OBJ.o 2>&1 | grep -A12 "Function Name: _ZN3dsr9k_viterbi\|error" | grep -E "error|SGPRs Spill|VGPRs Spill|ScratchSize" | head -n 3 | tr '\n' ' '; echo " <- $n";
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/var/$n/libdsr_hip.so lib/var/$n/$OBJ.o $OTHERS -Wl,-rpath,/opt/rocm/lib ) &
done
wait
