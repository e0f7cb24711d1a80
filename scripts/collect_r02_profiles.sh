#!/bin/bash
# Round-2 rocprofv3 evidence, run on the GPU box from the repo root (gpurun): kernel traces and PMC passes of the SHIPPED kernels.
# Counters in their own runs (no trace domains besides --kernel-trace), as MI355X_MICROARCH.md prescribes: SQ set | FETCH_SIZE | WRITE_SIZE.
# usage: bash scripts/collect_r02_profiles.sh [what ...]   what in: bench mh encode hmc fit egm bnn bnnmh   (default: all)
set -u
OUT=gpurun_out/r02prof
mkdir -p $OUT
export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
WHAT="${*:-bench mh encode hmc fit}"
summ() { for db in $(find $1 -name "*_results.db" 2>/dev/null); do python scripts/prof_summary.py $db; done; }
pmc3() {   # name, command...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc $SQ -d $OUT/${name}_sq -o sq -- "$@" > $OUT/${name}_sq.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/${name}_fetch -o fetch -- "$@" > $OUT/${name}_fetch.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${name}_write -o write -- "$@" > $OUT/${name}_write.log 2>&1
  { echo "# command: $*"; echo "# pass 1: --pmc $SQ"; summ $OUT/${name}_sq; echo "# pass 2: --pmc FETCH_SIZE GRBM_GUI_ACTIVE"; summ $OUT/${name}_fetch;
    echo "# pass 3: --pmc WRITE_SIZE"; summ $OUT/${name}_write; } > $OUT/r02_pmc_${name}.txt
}
for w in $WHAT; do
  case $w in
    bench)
      python bench.py --steps 2 --warmup 1 > $OUT/r02_bench_N1e6_1gpu.json 2> $OUT/bench.err
      timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_bench -o kt -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bayesian --no-fit --no-accuracy > $OUT/kt_bench.log 2>&1
      { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bayesian --no-fit --no-accuracy"; summ $OUT/kt_bench; } > $OUT/r02_kernel_trace_bench_N1e6.txt ;;
    mh) pmc3 mh_N1e6_60burn_40keep python scripts/probe_mh.py 1e6 100 40 ;;
    encode)
      python scripts/probe_encode.py 1e6 200 20 > $OUT/r02_encode_N1e6.json 2> $OUT/encode.err
      pmc3 encode_N1e6 python scripts/probe_encode.py 1e6 200 5 ;;
    hmc) pmc3 bgm_hmc python scripts/probe_bgm_wide.py 200000 10 ;;
    fit) pmc3 fit_B65536 python scripts/probe_fit.py 1e6 65536 dense 20 ;;
    egm) pmc3 egm_chain python scripts/probe_egm_native.py 200 200 ;;
    bnn) pmc3 bnn_chain env BNN_PROBE_SKIP_MH=1 python scripts/probe_bnn.py 20000 200 5 ;;
    bnnmh)   # sampling side of the Bayesian networks at the north-star panel size: MH iterations (burn-in, kept with 20 doses)
      timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_bnnmh -o kt -- env BNN_PROBE_SAMPLING_ONLY=1 python scripts/probe_bnn.py 1000000 200 10 > $OUT/kt_bnnmh.log 2>&1
      { echo "# rocprofv3 --kernel-trace --stats -- env BNN_PROBE_SAMPLING_ONLY=1 python scripts/probe_bnn.py 1000000 200 10"; grep -E "^MH|^predict" $OUT/kt_bnnmh.log | sed 's/^/# /'; summ $OUT/kt_bnnmh; } > $OUT/r02_kernel_trace_bnn_sampling_N1e6.txt
      pmc3 bnn_sampling_N1e6 env BNN_PROBE_SAMPLING_ONLY=1 python scripts/probe_bnn.py 1000000 200 5 ;;
  esac
done
ls -la $OUT | head -40
