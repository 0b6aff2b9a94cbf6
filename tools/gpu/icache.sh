#!/bin/bash
# instruction-cache counters of the iteration's kernels (single chain)
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAIT_ANY\|SQ_ACTIVE_INST_ANY\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES" | sort -u > $O/icache_counters.txt
cat $O/icache_counters.txt
rm -rf /tmp/ic1 /tmp/ic2
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d /tmp/ic1 -o r -f csv -- python3 $GRAFT_REPO_ROOT/tools/prof_workload.py --workload warm --chains 1 --steps 60 > $O/ic1.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/sq_counters.py /tmp/ic1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH --kernel-trace -d /tmp/ic2 -o r -f csv -- python3 $GRAFT_REPO_ROOT/tools/prof_workload.py --workload warm --chains 1 --steps 60 > $O/ic2.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/sq_counters.py /tmp/ic2
tail -3 $O/ic1.log $O/ic2.log
