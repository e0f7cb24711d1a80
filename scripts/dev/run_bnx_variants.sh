# dev aid: time the split-precision sampler of every library variant named on the command line (scripts/dev/build_bnx_variant.sh)
for n in "$@"; do
  echo "== $n"; BGM_HIP_LIB=$PWD/bayesgm_amd/csrc/build/var/lib_$n.so BNN_PRECISION=f16x3 BNN_PROBE_SAMPLING_ONLY=1 python scripts/probe_bnn.py 1000000 200 10 2>&1 | grep "^MH"
done
