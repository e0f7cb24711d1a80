#!/bin/bash
# rocprofv3 evidence for the fixed-normalisation Bayesian-network sampling kernels (bnf_*), run on the GPU box from the repo root.
# usage: bash scripts/prof_bnf.sh <tag> [kt] [sq] [sq2] [mem]     environment (BGM_BNF_CFG, ...) is inherited by the probe
# PROF_CMD="..." profiles another command with the same passes (wide HMC: "python scripts/probe_bgm_wide.py 196608 5"; the
# deterministic sampler's keep phase: "python scripts/probe_mh.py 1e6 40 40")
set -u
TAG=${1:-bnf}; shift
WHAT="${*:-kt sq}"
OUT=gpurun_out/r03prof
mkdir -p $OUT
export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"
CMD=${PROF_CMD:-"env BNN_PROBE_SAMPLING_ONLY=1 python scripts/probe_bnn.py 1000000 200 5"}
summ() { for db in $(find $1 -name "*_results.db" 2>/dev/null); do python scripts/prof_summary.py $db; done; }
for w in $WHAT; do
  case $w in
    kt) timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt -- $CMD > $OUT/${TAG}_kt.log 2>&1
        { echo "# rocprofv3 --kernel-trace --stats -- $CMD   (BGM_BNF_CFG=${BGM_BNF_CFG:-default} BGM_BNF_ECFG=${BGM_BNF_ECFG:-default})"; grep -E "^MH|^predict" $OUT/${TAG}_kt.log | sed 's/^/# /'; summ $OUT/${TAG}_kt; } > $OUT/r03_kernel_trace_${TAG}.txt ;;
    sq) timeout 600 rocprofv3 --kernel-trace --pmc $SQ -d $OUT/${TAG}_sq -o sq -- $CMD > $OUT/${TAG}_sq.log 2>&1
        { echo "# rocprofv3 --kernel-trace --pmc $SQ -- $CMD   (BGM_BNF_CFG=${BGM_BNF_CFG:-default} BGM_BNF_ECFG=${BGM_BNF_ECFG:-default})"; summ $OUT/${TAG}_sq; } > $OUT/r03_pmc_sq_${TAG}.txt ;;
    sq2) timeout 600 rocprofv3 --kernel-trace --pmc $SQ2 -d $OUT/${TAG}_sq2 -o sq2 -- $CMD > $OUT/${TAG}_sq2.log 2>&1
        { echo "# rocprofv3 --kernel-trace --pmc $SQ2 -- $CMD"; summ $OUT/${TAG}_sq2; } > $OUT/r03_pmc_sq2_${TAG}.txt ;;
    mem) timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/${TAG}_fetch -o fetch -- $CMD > $OUT/${TAG}_fetch.log 2>&1
         timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o write -- $CMD > $OUT/${TAG}_write.log 2>&1
         timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $OUT/${TAG}_tcc -o tcc -- $CMD > $OUT/${TAG}_tcc.log 2>&1
         { echo "# --pmc FETCH_SIZE GRBM_GUI_ACTIVE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -- $CMD"; summ $OUT/${TAG}_fetch; summ $OUT/${TAG}_write; summ $OUT/${TAG}_tcc; } > $OUT/r03_pmc_mem_${TAG}.txt ;;
  esac
done
# the raw rocprofv3 databases are tens of MB each and gpurun carries at most 64 MiB back: keep the summaries and logs only
find $OUT -mindepth 1 -maxdepth 1 -type d -name "${TAG}_*" -exec rm -rf {} +
