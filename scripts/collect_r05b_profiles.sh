#!/bin/bash
# round-5 (second half) evidence, run on the GPU box from the repo root: split-precision Bayesian sampler (bnx) and general-width engine phases
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05b
# kernel trace + SQ / memory counters of the fp16x3 Bayesian sampler at the bench shape (prof_bnf.sh writes gpurun_out/r03prof/r03_*_<tag>.txt)
PROF_CMD="env BNN_PRECISION=f16x3 BNN_PROBE_SAMPLING_ONLY=1 python scripts/probe_bnn.py 1000000 200 5" bash scripts/prof_bnf.sh bnx kt sq sq2 mem
for f in kernel_trace pmc_sq pmc_sq2 pmc_mem; do cp gpurun_out/r03prof/r03_${f}_bnx.txt gpurun_out/r05b/r05_${f}_bnn_sampling_f16x3_N1e6.txt; done
# per-phase cycle stamps of the general-width engine (development build with -D GX_PHASE_CLOCK) and its counters at the forced default widths
{ echo "# BGM_FORCE_GX=1 python scripts/probe_gx.py with a -D GX_PHASE_CLOCK build of gx_api.hip: cycles of thread 0 of every workgroup per phase of a transition"
  BGM_HIP_LIB=$PWD/bayesgm_amd/csrc/build/var/lib_gxclk.so BGM_FORCE_GX=1 python scripts/probe_gx.py 2>&1 | grep "GX_PHASE\|^gx  "
  echo "# product build:"; BGM_FORCE_GX=1 python scripts/probe_gx.py 2>&1 | grep "^gx  "; } > gpurun_out/r05b/r05_gx_phases.txt
PROF_CMD="env BGM_FORCE_GX=1 GX_ONLY=default python scripts/probe_gx.py" bash scripts/prof_bnf.sh gx_default sq sq2
cp gpurun_out/r03prof/r03_pmc_sq_gx_default.txt gpurun_out/r05b/r05_pmc_sq_gx_default_widths.txt
cp gpurun_out/r03prof/r03_pmc_sq2_gx_default.txt gpurun_out/r05b/r05_pmc_sq2_gx_default_widths.txt
