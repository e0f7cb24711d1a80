#!/bin/bash
# dev aid: libbgm_hip.so variant that differs in ONE translation unit's flags: scripts/dev/build_variant.sh NAME unit.hip [extra flags...]
cd "$(dirname "$0")/../../bayesgm_amd/csrc" || exit 1
n=$1; unit=$2; shift 2
d=build/var/$n; mkdir -p $d
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result -I ."
case $unit in bnf_api.hip|bnx_api.hip|bnf_det_api.hip|causal_bx3_api.hip) F="$F -fno-slp-vectorize";; esac
OBJS=$(ls build/libbgm_hip.so.*.o | grep -v "\.$unit\.o")
hipcc $F "$@" -c $unit -o $d/unit.o -save-temps=obj 2>$d/err.txt || { grep -m5 error $d/err.txt; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/lib_$n.so $OBJS $d/unit.o
rm -f $d/*.bc $d/*.hipi $d/*.out $d/*.hipfb $d/*host*
echo build/var/lib_$n.so
