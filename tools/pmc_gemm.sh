#!/bin/bash
# What do the trailing-update kernels wait for? (VERDICT r04 "Next 6") Counter passes over bench.py's default workload (and, with a second argument,
# another one), --kernel-trace + --pmc only, a few counters per pass (SQ counters share one block: up to 8 per pass):
#   SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (issuing), quad-cycles
#   SQ_ACTIVE_INST_LDS / SQ_INSTS_LDS / SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: LDS pressure; SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES: matrix pipe
# usage (on the GPU box): bash tools/pmc_gemm.sh <tag> [workload]
tag=${1:-pmcg}; wl=${2:-mh12345}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pmcg_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcg_$i -o pmc -- python $root/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --sustain-s 0 --a12-leg 0 > $root/gpurun_out/${tag}_pmcg_$i.log 2>&1
  db=$(ls /tmp/pmcg_$i/*.db 2>/dev/null | head -1)
  if [[ -n "$db" ]]; then (cd $root; python tools/rocpd_counters.py $db gpurun_out/${tag}_pmc_gemm_${wl}_set$i.csv | grep -E "^Name|k_gemm_abt|k_potrf|k_trsm" | cut -c1-400); else echo "set $i: no database (counter unknown?)"; tail -3 $root/gpurun_out/${tag}_pmcg_$i.log; fi
done
