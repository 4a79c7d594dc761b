#!/bin/bash
# Round 6, GPU call 18: stall attribution of k_step from the counters: average latency of a vector-memory and of an
# LDS instruction (level / count), L2 round trip as the L1 sees it, the texture path's stalls, instruction-cache misses, the L2's
# write-request stalls - one rocprofv3 --pmc pass each over 100 steady steps of the bench's step (counters never combined with traces).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step17
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras --no-pmc"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_LEVEL_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_ANY" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_STALL SQ_INSTS_SMEM" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pass $i rc=$?"
done
python $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
find $OUT \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
grep -A 12 "k_step<512, false>" $OUT/summary.txt | grep -E "mean=|counters|k_step" | cut -c1-120
