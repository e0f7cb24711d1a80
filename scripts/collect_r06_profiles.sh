#!/bin/bash
# Round-6 rocprofv3 evidence, run on the GPU box from the repo root (gpurun).  Counters in their own passes (--kernel-trace only beside
# --pmc), as MI355X_MICROARCH.md prescribes.  Summaries land in gpurun_out/r06prof/ and are copied to profiles/ by hand.
# usage: bash scripts/collect_r06_profiles.sh [what ...]   what in: mh bnn bnw hmc hmcb hmcmem c1 hmcw bench ablation    (default: mh bnn)
set -u
OUT=gpurun_out/r06prof
mkdir -p $OUT
export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
WHAT="${*:-mh bnn}"
summ() { for db in $(find $1 -name "*_results.db" 2>/dev/null); do python scripts/prof_summary.py $db; done; }
passes() {   # tag, command...
  local tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${tag}_kt -o kt -- "$@" > $OUT/${tag}_kt.log 2>&1 < /dev/null
  { echo "# rocprofv3 --kernel-trace --stats -- $*"; grep -E "^(False|wave|True|max|N=|\{)" $OUT/${tag}_kt.log | sed 's/^/# /'; summ $OUT/${tag}_kt; } > $OUT/r06_kernel_trace_${tag}.txt
  timeout 900 rocprofv3 --kernel-trace --pmc $SQ -d $OUT/${tag}_sq -o sq -- "$@" > $OUT/${tag}_sq.log 2>&1 < /dev/null
  { echo "# rocprofv3 --kernel-trace --pmc $SQ -- $*"; summ $OUT/${tag}_sq; } > $OUT/r06_pmc_sq_${tag}.txt
  timeout 900 rocprofv3 --kernel-trace --pmc $SQ2 -d $OUT/${tag}_sq2 -o sq2 -- "$@" > $OUT/${tag}_sq2.log 2>&1 < /dev/null
  { echo "# rocprofv3 --kernel-trace --pmc $SQ2 -- $*"; summ $OUT/${tag}_sq2; } > $OUT/r06_pmc_sq2_${tag}.txt
  find $OUT -mindepth 1 -maxdepth 1 -type d -name "${tag}_*" -exec rm -rf {} +
}
mempasses() {   # tag, command...: HBM bytes (FETCH_SIZE and WRITE_SIZE need a pass each: MI355X_MICROARCH.md, rocprofv3 PMC slots)
  local tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/${tag}_fetch -o fetch -- "$@" > $OUT/${tag}_fetch.log 2>&1 < /dev/null
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${tag}_write -o write -- "$@" > $OUT/${tag}_write.log 2>&1 < /dev/null
  { echo "# --pmc FETCH_SIZE GRBM_GUI_ACTIVE | WRITE_SIZE (one pass each) -- $*"; summ $OUT/${tag}_fetch; summ $OUT/${tag}_write; } > $OUT/r06_pmc_mem_${tag}.txt
  find $OUT -mindepth 1 -maxdepth 1 -type d -name "${tag}_*" -exec rm -rf {} +
}
for w in $WHAT; do
  case $w in
    # HBM traffic of the split-precision BGM kernels at C4's shape (the data rows are re-read per gradient evaluation: by design, and counted)
    hmcmem) mempasses bgm_hmc_f16x3 env BGM_PROBE_PRECISION=f16x3 python scripts/probe_bgm_wide.py 2e5 4
            mempasses bgmf_hmc_f16x3 env BGM_PROBE_MODES=f16x3 python scripts/probe_bgmf.py 2e5 4 ;;
    # the pure-transition kernel that is now 4.5 of the product predict's 4.9 s: 100 burn-in + 40 kept iterations
    mh) passes causal_mh python scripts/probe_mh.py 1e6 100 40 ;;
    # the Bayesian default model's sampler + effects kernels at N = 1e6 (item 1)
    bnn) passes bnn_sampling_N1e6 env BNN_PROBE_SAMPLING_ONLY=1 python scripts/probe_bnn.py 1000000 200 5 ;;
    # wide Bayesian nets (item 7: a tracked number for bnw_*)
    bnw) passes bnw_wide python scripts/probe_bnw.py 1e5 5 ;;
    # BGM HMC at C4's shape, fp32 and split-precision heads
    hmc) passes bgm_hmc_f16x3 env BGM_PROBE_PRECISION=f16x3 python scripts/probe_bgm_wide.py 2e5 4 ;;
    # the Bayesian generator's frozen-noise HMC at C4's shape in split precision (bgmfx_hmc_kernel)
    hmcb) passes bgmf_hmc_f16x3 env BGM_PROBE_MODES=f16x3 python scripts/probe_bgmf.py 2e5 4
          python scripts/probe_bgmf.py 2e5 5 > $OUT/r06_bgmf_hmc_fp32_vs_f16x3.txt 2>/dev/null ;;
    # configs[1] through the class (event form of the binary-treatment retained phase): kernel trace only
    c1)
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/c1_kt -o kt -- python scripts/probe_c1.py > $OUT/c1_kt.log 2>&1 < /dev/null
      { echo "# rocprofv3 --kernel-trace --stats -- python scripts/probe_c1.py"; grep -E "^\{" $OUT/c1_kt.log | sed 's/^/# /'; summ $OUT/c1_kt; } > $OUT/r06_kernel_trace_config_c1.txt
      rm -rf $OUT/c1_kt ;;
    # split-precision heads: waves per block
    hmcw)
      for w in 8 12 16; do echo "BGM_X3_WAVES=$w"; BGM_X3_WAVES=$w BGM_PROBE_PRECISION=f16x3 python scripts/probe_bgm_wide.py 2e5 4; done > $OUT/r06_bgm_hmc_f16x3_waves.txt 2>&1
      python scripts/probe_bgm_wide.py 2e5 4 >> $OUT/r06_bgm_hmc_f16x3_waves.txt 2>&1 ;;
    bench)
      timeout 1500 rocprofv3 --kernel-trace --stats -d $OUT/kt_bench -o kt -- python bench.py > $OUT/kt_bench.log 2>&1 < /dev/null
      { echo "# rocprofv3 --kernel-trace --stats -- python bench.py   (the default invocation)"; summ $OUT/kt_bench; } > $OUT/r06_kernel_trace_bench_N1e6.txt
      grep -E "^\{" $OUT/kt_bench.log | tail -1 > $OUT/r06_bench_N1e6_1gpu_under_rocprof.json
      rm -rf $OUT/kt_bench ;;
    # what the Rademacher sign applications / all epilogues of the Flipout layers cost (development builds, csrc/build/abl/)
    ablation)
      for v in product nosign noepi; do
        lib=bayesgm_amd/csrc/build/abl/lib_$v.so
        [ $v = product ] && lib=bayesgm_amd/libbgm_hip.so
        [ -f $lib ] || continue
        BGM_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 1 --warmup 0 --burn-in 500 --n-mcmc 300 --no-cpu-baseline --no-fit --no-bgm --no-general-width --no-bf16x3 --no-accuracy --no-end-to-end --no-configs 2> $OUT/abl_$v.err | python -c "
import json, sys
d = json.loads(sys.stdin.readline())['bayesian_nets']
print('$v', json.dumps({k: d[k] for k in ('burn_in_iteration_ms', 'kept_iteration_ms', 'sampler_frac_of_fp32_mfma_peak', 'effects_frac_of_fp32_mfma_peak', 'value')}))" >> $OUT/r06_bnf_ablation.txt
      done ;;
  esac
done
ls -la $OUT
