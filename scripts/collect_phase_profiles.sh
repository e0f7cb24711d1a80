#!/bin/bash
# Phase profiles of the general Bayesian kernels (development builds with shader-clock stamps of wave 0).  Build first, here:
#   bash scripts/build_variant.sh bnwprof "-DBNW_PROF" bnw_api.hip; bash scripts/build_variant.sh bnnprof "-DBNN_PROF" bnn_api.hip
# then on the GPU box: bash scripts/collect_phase_profiles.sh   -> gpurun_out/r06prof/r06_phase_profiles.txt
mkdir -p gpurun_out/r06prof
{
  echo "# BGM_HIP_LIB=ab/lib_bnwprof.so python scripts/probe_bnw.py 1e5 5     (the any-width Bayesian sampling path, final build)"
  BGM_HIP_LIB=$PWD/ab/lib_bnwprof.so timeout 200 python scripts/probe_bnw.py 1e5 5 2>&1 | grep -E "^bnw|BNW_PROF"
  echo "# BGM_BNN_STEP_ONE_LAUNCH=1 BGM_HIP_LIB=ab/lib_bnnprof.so python scripts/probe_bnn_fit_wide.py 20000 100     (the one-launch theta step of g: where its time went)"
  BGM_BNN_STEP_ONE_LAUNCH=1 BGM_HIP_LIB=$PWD/ab/lib_bnnprof.so timeout 300 python scripts/probe_bnn_fit_wide.py 20000 100 2>&1 | grep -E "^use_bnn|BNN_PROF" | grep -v default
  echo "# BGM_HIP_LIB=ab/lib_bnnprof.so python scripts/probe_bnn_fit_wide.py 20000 100     (the step kernel of g with the launches over the chip around it)"
  BGM_HIP_LIB=$PWD/ab/lib_bnnprof.so timeout 300 python scripts/probe_bnn_fit_wide.py 20000 100 2>&1 | grep -E "^use_bnn|BNN_PROF" | grep -v default
} > gpurun_out/r06prof/r06_phase_profiles.txt 2>&1
cut -c1-260 gpurun_out/r06prof/r06_phase_profiles.txt
