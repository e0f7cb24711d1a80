#!/bin/bash
# Development A/B builds: recompile the named translation units with extra -D flags and link them with the product objects of the
# other units (bayesgm_amd/csrc/build/) into ab/lib_<tag>.so; run with BGM_HIP_LIB=$PWD/ab/lib_<tag>.so.
# usage: bash scripts/build_variant.sh <tag> "<-D flags>" unit.hip [unit.hip ...]        (python -m bayesgm_amd.csrc.build first)
set -eu
tag=$1; defs=$2; shift 2
B=bayesgm_amd/csrc/build
mkdir -p ab
objs=$(ls $B/libbgm_hip.so.*.hip.o)
for u in "$@"; do
  extra=""
  case $u in bnf_api.hip|bnx_api.hip|bnf_det_api.hip|causal_bx3_api.hip) extra="-fno-slp-vectorize" ;; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result $extra $defs \
    -I bayesgm_amd/csrc -c bayesgm_amd/csrc/$u -o ab/$tag.$u.o &
done
wait
for u in "$@"; do objs=$(echo "$objs" | grep -v "/libbgm_hip.so.$u.o\$"); objs="$objs ab/$tag.$u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_$tag.so $objs
ls -la ab/lib_$tag.so
