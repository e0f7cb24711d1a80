#!/bin/bash
# Round-4 rocprofv3 evidence, run on the GPU box from the repo root (gpurun).  Counters in their own passes (--kernel-trace only beside
# --pmc), as MI355X_MICROARCH.md prescribes.  Summaries land in gpurun_out/r04prof/ and are copied to profiles/ by hand.
# usage: bash scripts/collect_r04_profiles.sh [what ...]   what in: bench keep bx3 hx3 gx bvn bvnc bvnf fit   (default: all)
set -u
OUT=gpurun_out/r04prof
mkdir -p $OUT
export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
WHAT="${*:-bench keep gx bvn fit}"
summ() { for db in $(find $1 -name "*_results.db" 2>/dev/null); do python scripts/prof_summary.py $db; done; }
passes() {   # tag, command...
  local tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${tag}_kt -o kt -- "$@" > $OUT/${tag}_kt.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- $*"; grep -E "^(gx|resident|\{|N=|  library)" $OUT/${tag}_kt.log | sed 's/^/# /'; summ $OUT/${tag}_kt; } > $OUT/r04_kernel_trace_${tag}.txt
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ -d $OUT/${tag}_sq -o sq -- "$@" > $OUT/${tag}_sq.log 2>&1
  { echo "# rocprofv3 --kernel-trace --pmc $SQ -- $*"; summ $OUT/${tag}_sq; } > $OUT/r04_pmc_sq_${tag}.txt
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/${tag}_fetch -o fetch -- "$@" > $OUT/${tag}_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${tag}_write -o write -- "$@" > $OUT/${tag}_write.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $OUT/${tag}_tcc -o tcc -- "$@" > $OUT/${tag}_tcc.log 2>&1
  { echo "# --pmc FETCH_SIZE GRBM_GUI_ACTIVE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum (one pass each) -- $*"; summ $OUT/${tag}_fetch; summ $OUT/${tag}_write; summ $OUT/${tag}_tcc; } > $OUT/r04_pmc_mem_${tag}.txt
  find $OUT -mindepth 1 -maxdepth 1 -type d -name "${tag}_*" -exec rm -rf {} +
}
for w in $WHAT; do
  case $w in
    bench)
      python bench.py --steps 2 --warmup 1 > $OUT/r04_bench_N1e6_1gpu.json 2> $OUT/bench.err
      timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_bench -o kt -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bayesian --no-fit --no-accuracy --no-bgm > $OUT/kt_bench.log 2>&1
      { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bayesian --no-fit --no-accuracy --no-bgm"; summ $OUT/kt_bench; } > $OUT/r04_kernel_trace_bench_N1e6.txt
      rm -rf $OUT/kt_bench ;;
    keep) passes causal_mh python scripts/probe_mh.py 1e6 100 40 ;;
    bx3) passes causal_mh_bf16x3 python scripts/probe_mh.py 1e6 100 40 bf16x3 ;;
    hx3) passes causal_mh_f16x3 python scripts/probe_mh.py 1e6 100 40 f16x3 ;;
    gx) passes gx_w256 env BGM_FORCE_GX=1 GX_ONLY=w256 python scripts/probe_gx.py 250000 ;;
    bvn) passes bvn_hmc_frozen env BGM_BVN_NO_CHAINS=1 python scripts/probe_bvn_hmc.py 400000 3 ;;      # the LDS-tile engine (gxf_bgm_hmc_kernel)
    bvnf) passes bvn_hmc_fresh python scripts/probe_bvn_hmc.py 400000 3 fresh ;;                        # row-tile chains, fresh noise (linear stream)
    bvnc) passes bvn_hmc_chains python scripts/probe_bvn_hmc.py 400000 3 ;;                            # row-tile chains (bgmf_hmc_kernel)
    fit) timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/fit_kt -o kt -- python scripts/probe_fit.py 1e6 32 replay 2000 > $OUT/fit_kt.log 2>&1
         { echo "# rocprofv3 --kernel-trace --stats -- python scripts/probe_fit.py 1e6 32 replay 2000   (2005 minibatches from Python + 2005 from bgm_causal_fit_epoch)"; grep -E "^(N=|  library)" $OUT/fit_kt.log | sed 's/^/# /'; summ $OUT/fit_kt; } > $OUT/r04_kernel_trace_fit_B32.txt
         rm -rf $OUT/fit_kt ;;
  esac
done
ls -la $OUT | head -40
