#!/bin/bash
# usage: tools/build_gmm_variants.sh name1 "-DFLAGS" name2 "-DFLAGS" ...   -> distantspeechrecognition-mirror_amd/lib/var/<name>/libdsr_hip.so
cd /root/repo/distantspeechrecognition-mirror_amd
OTHERS=$(ls lib/*.o | grep -v k_gmm_mfma.o)
while [ $# -gt 1 ]; do
  n=$1; fl=$2; shift 2; mkdir -p lib/var/$n
  ( /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -I../include -ffp-contract=off $fl -c csrc/k_gmm_mfma.hip -o lib/var/$n/k_gmm_mfma.o 2>&1 | grep -v warning;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/var/$n/libdsr_hip.so lib/var/$n/k_gmm_mfma.o $OTHERS -Wl,-rpath,/opt/rocm/lib; echo "built $n" ) &
done
wait
