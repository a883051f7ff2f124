#!/bin/bash
# Collect the rocprofv3 evidence for one round.  Usage (on the GPU box, via gpurun):
#   bash profiles/run_profiles.sh r01
# Writes raw output under gpurun_out/prof_<round>/ ; copy the summaries you want judged into profiles/.
# PMC counters are collected in their own passes, never combined with other trace domains.
# Extra arguments go to bench.py:  bash profiles/run_profiles.sh r02_bf16 --dtype bf16
R=${1:-r01}
[ $# -gt 0 ] && shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
CMD=(python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@")
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- "${CMD[@]}" > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_sq -- "${CMD[@]}" > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- "${CMD[@]}" > $OUT/pmc_sq2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- "${CMD[@]}" > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- "${CMD[@]}" > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tcc -- "${CMD[@]}" > $OUT/pmc_tcc.log 2>&1
# (round 6) how many of the L2's memory-side requests are addressed to this device's DRAM (the others: IO / GMI) -- the Infinity Cache sits in
# front of the memory controller, these counters do not tell its hits from misses
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/pmc_tcc2 -- "${CMD[@]}" > $OUT/pmc_tcc2.log 2>&1
find $OUT -name "*.csv" | head -50
