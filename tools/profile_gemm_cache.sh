#!/bin/bash
# Where do the operand fills of the tiled GEMM come from?  L2 hit / miss / request counters and the vector-cache stall counters of
# pswin_gemm_nt on one shape, one rocprofv3 --pmc pass per group (program directly after --; no trace domains beside the counters).
# usage (GPU box): bash tools/profile_gemm_cache.sh M K N tile_m   -> gpurun_out/gemm_cache_*.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M=${1:-16384}; K=${2:-384}; N=${3:-1536}; T=${4:-128}
O=gpurun_out/gemm_cache; rm -rf $O; mkdir -p $O
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python3 tools/pmc_gemm_nt.py $M $K $N $T > /dev/null 2> $O/p$i.err
  f=$(ls $O/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_rows.py $f gemm_nt >> gpurun_out/gemm_cache_${M}_${K}_${N}.txt; else echo "group '$grp': no output ($(tail -1 $O/p$i.err | cut -c1-200))" >> gpurun_out/gemm_cache_${M}_${K}_${N}.txt; fi
done
rm -rf $O
cat gpurun_out/gemm_cache_${M}_${K}_${N}.txt
