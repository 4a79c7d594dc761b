#!/bin/bash
# Round 6, GPU call 17: which counters this pool's rocprofv3 offers for a stall attribution of k_step (vector-memory / texture path,
# L2 request stalls, instruction fetch), and one pass over the driver-form command with the ones that exist.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || rocprofv3 -L > $OUT/avail.txt 2>&1
grep -oE "\b(SQ_[A-Z_0-9]+|TA_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|TCC_[A-Z_0-9a-z]+|TD_[A-Z_0-9a-z]+|SQC_[A-Z_0-9a-z]+|GRBM_[A-Z_0-9a-z]+)\b" $OUT/avail.txt | sort -u > $OUT/counter_names.txt
wc -l $OUT/counter_names.txt
grep -E "^SQ_(WAIT|INST_CYCLES|ACTIVE_INST|IFETCH|INSTS_VMEM|INSTS_FLAT|INST_LEVEL|INSTS_SMEM|BUSY_CU|WAVES|VMEM|LEVEL)" $OUT/counter_names.txt | tr '\n' ' '; echo
grep -E "^(TA_|TCP_|SQC_)" $OUT/counter_names.txt | grep -iE "busy|stall|pending|miss|hit|req$|latency" | tr '\n' ' ' | cut -c1-1500; echo
grep -E "^TCC_" $OUT/counter_names.txt | grep -iE "stall|busy|req_sum|hit_sum|miss_sum|wrreq" | tr '\n' ' ' | cut -c1-1200; echo
rm -f $OUT/avail.txt.big; [ $(stat -c %s $OUT/avail.txt) -gt 4000000 ] && gzip $OUT/avail.txt
