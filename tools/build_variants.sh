#!/bin/bash
# usage: tools/build_variants.sh name1 "-DFLAGS" name2 "-DFLAGS" ...   -> distantspeechrecognition-mirror_amd/lib/var/<name>/libdsr_hip.so
# (a name of the form name=path/to/source.hip builds that source file as the decoder instead of csrc/k_viterbi.hip)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/distantspeechrecognition-mirror_amd
OTHERS=$(ls lib/*.o | grep -v k_viterbi.o)
while [ $# -gt 1 ]; do
  n=$1; fl=$2; shift 2; src=csrc/k_viterbi.hip
  case $n in *=*) src=${n#*=}; n=${n%%=*};; esac
  mkdir -p lib/var/$n
  ( /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -I../include -Icsrc -ffp-contract=off $fl -Rpass-analysis=kernel-resource-usage -c $src -o lib/var/$n/k_viterbi.o 2>&1 | grep -A12 "Function Name: _ZN3dsr9k_viterbi" | grep -E "SGPRs Spill|VGPRs Spill|ScratchSize" | head -n 3 | tr '\n' ' '; echo " <- $n";
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/var/$n/libdsr_hip.so lib/var/$n/k_viterbi.o $OTHERS -Wl,-rpath,/opt/rocm/lib ) &
done
wait
