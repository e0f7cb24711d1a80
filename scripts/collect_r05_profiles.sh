#!/bin/bash
# Round-5 rocprofv3 evidence, run on the GPU box from the repo root (gpurun).  Counters in their own passes (--kernel-trace only beside
# --pmc), as MI355X_MICROARCH.md prescribes.  Summaries land in gpurun_out/r05prof/ and are copied to profiles/ by hand.
# usage: bash scripts/collect_r05_profiles.sh [what ...]   what in: bench event    (default: all)
set -u
OUT=gpurun_out/r05prof
mkdir -p $OUT
export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
WHAT="${*:-bench event}"
summ() { for db in $(find $1 -name "*_results.db" 2>/dev/null); do python scripts/prof_summary.py $db; done; }
passes() {   # tag, command...
  local tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${tag}_kt -o kt -- "$@" > $OUT/${tag}_kt.log 2>&1 < /dev/null
  { echo "# rocprofv3 --kernel-trace --stats -- $*"; grep -E "^(False|wave|True|max|N=)" $OUT/${tag}_kt.log | sed 's/^/# /'; summ $OUT/${tag}_kt; } > $OUT/r05_kernel_trace_${tag}.txt
  timeout 900 rocprofv3 --kernel-trace --pmc $SQ -d $OUT/${tag}_sq -o sq -- "$@" > $OUT/${tag}_sq.log 2>&1 < /dev/null
  { echo "# rocprofv3 --kernel-trace --pmc $SQ -- $*"; summ $OUT/${tag}_sq; } > $OUT/r05_pmc_sq_${tag}.txt
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/${tag}_fetch -o fetch -- "$@" > $OUT/${tag}_fetch.log 2>&1 < /dev/null
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${tag}_write -o write -- "$@" > $OUT/${tag}_write.log 2>&1 < /dev/null
  timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $OUT/${tag}_tcc -o tcc -- "$@" > $OUT/${tag}_tcc.log 2>&1 < /dev/null
  { echo "# --pmc FETCH_SIZE GRBM_GUI_ACTIVE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum (one pass each) -- $*"; summ $OUT/${tag}_fetch; summ $OUT/${tag}_write; summ $OUT/${tag}_tcc; } > $OUT/r05_pmc_mem_${tag}.txt
  find $OUT -mindepth 1 -maxdepth 1 -type d -name "${tag}_*" -exec rm -rf {} +
}
for w in $WHAT; do
  case $w in
    bench)
      timeout 1500 python bench.py --steps 2 --warmup 1 > $OUT/r05_bench_N1e6_1gpu.json 2> $OUT/bench.err < /dev/null
      timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_bench -o kt -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bayesian --no-fit --no-accuracy --no-bgm --no-end-to-end --no-bf16x3 > $OUT/kt_bench.log 2>&1 < /dev/null
      { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bayesian --no-fit --no-accuracy --no-bgm --no-end-to-end --no-bf16x3   (one headline predict with the outcome cache off, one per chain = event form, one per wave)"; summ $OUT/kt_bench; } > $OUT/r05_kernel_trace_bench_N1e6.txt
      rm -rf $OUT/kt_bench ;;
    # the retained phase at the bench shape: 60 burn-in + 82 retained iterations (two segments of the event form at the default budget),
    # outcome cache off / per wave / per chain (event form) twice
    event) passes event_form env BURN=60 KEEP=82 python scripts/probe_event.py 1e6 ;;
  esac
done
ls -la $OUT
