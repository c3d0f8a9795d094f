#!/bin/bash
# build the working tree's kernels into exp_libs/libcagym_<tag>.so (A/B experiments; extra flags after the tag)
set -e
T=$1; shift
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
FL="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -mllvm -disable-machine-licm"
mkdir -p exp_libs
hipcc --offload-arch=gfx950 -c -DCAGYM_MONOLITHIC $FL "$@" -o /tmp/alt_$T.o gym-exploration-2d_amd/csrc/cagym_api.hip
g++ -shared -o exp_libs/libcagym_$T.so /tmp/alt_$T.o -L$TL -lamdhip64 -Wl,-rpath,$TL
echo exp_libs/libcagym_$T.so
