#!/bin/bash
# dev aid: build libbgm_hip.so variants that differ in the split-precision sampler's (R, W) only: scripts/dev/build_bnx_variant.sh R W ER EW [extra -D...]
cd "$(dirname "$0")/../../bayesgm_amd/csrc" || exit 1
n=r$1w$2e$3w$4$6; d=build/var/$n; mkdir -p $d
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result -fno-slp-vectorize -I ."
OBJS=$(ls build/libbgm_hip.so.*.o | grep -v "bnx_api.hip.o")
hipcc $F -DBNX_R=$1 -DBNX_W=$2 -DBNX_ER=$3 -DBNX_EW=$4 $5 -c bnx_api.hip -o $d/bnx.o -save-temps=obj 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/lib_$n.so $OBJS $d/bnx.o
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|name):" $d/bnx_api-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - | grep "bnf_mh_kernelILi3.*Li1ELb0\|bnf_effects_kernelILi1" | awk -v n=$n '{print n, $2, $4, $6}'
rm -f $d/*.bc $d/*.hipi $d/*.out $d/*.hipfb $d/*host*
